import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, time
import cases, oracle, parity
from fabber_core_amd import hiplib
from fabber_core_amd.device import DeviceProblem
V=1_000_000
h,y = cases.exp_problem(V,100,2,0.02,seed=20260103,max_iterations=50)
prob = DeviceProblem(h,y,'cuda:0')
def timeit():
    prob.run(); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record(); prob.run(); prob.run(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/2
# single-step error at k=8 state for 2000 voxels
Vs=2048
hs, ys = cases.exp_problem(Vs,100,2,0.02,seed=20260103,max_iterations=8)
state = oracle.run(hs, ys)['mvn']
h1,_ = cases.exp_problem(Vs,100,2,0.02,seed=20260103,max_iterations=1, init_mvn=state)
ref = oracle.run(h1, ys)
for mode,tol in (('moments',0),('auto',1e-12),('auto',1e-10),('auto',1e-8),('auto',1e-6),('auto',1e-4),('exact',0)):
    hiplib.set_residual_mode(mode); hiplib.set_residual_tolerance(tol if tol else 1e-6)
    ms = timeit()
    res = prob.results(); bad = int((res['status']!=0).sum())
    got = hiplib.run_host(h1, ys)
    ok = (ref['status']==0)&(got['status']==0)
    e,_,_ = parity.voxel_errors(h1, ref, got, ok)
    print('%-8s tol %-7g  %.2f ms  %.1f Mvox/s  bad %d | single-step(k=8) err q50 %.1e q90 %.1e q99 %.1e max %.1e' % (mode, tol, ms, V/ms/1e3, bad, *np.quantile(e,[.5,.9,.99]), e.max()))
