import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import cases, oracle, hipengine
from fabber_core_amd import hiplib, vbabi
np.set_printoptions(precision=4, linewidth=220)
def report(name, h, y):
    a = oracle.run(h, y); b = hipengine.run(h, y)
    P = h.cfg.n_params; n = P+1
    print('==', name, 'kernel', hiplib.kernel_name(h))
    print('  status oracle', np.bincount(a['status'], minlength=5), 'gpu', np.bincount(b['status'], minlength=5), 'mismatch', (a['status']!=b['status']).sum())
    print('  iter mismatch', (a['iterations']!=b['iterations']).sum(), 'iters oracle', np.bincount(a['iterations'])[:60].nonzero()[0])
    ok = (a['status']==0)&(b['status']==0)&(a['iterations']==b['iterations'])
    ca, ma = oracle.unpack_mvn(a['mvn'][:,ok], n); cb, mb = oracle.unpack_mvn(b['mvn'][:,ok], n)
    sd = np.sqrt(np.abs(np.einsum('vii->vi', ca)))
    em = np.abs(ma-mb)/np.maximum(np.abs(ma), sd); ec = np.abs(ca-cb)/(sd[:,:,None]*sd[:,None,:])
    print('  err means max', em.max(axis=0), ' cov max', ec.max(), 'relmeans', (np.abs(ma-mb)/np.maximum(np.abs(ma),1e-12))[:, :P].max())
    if h.cfg.need_f:
        Fa, Fb = a['free_energy'][ok], b['free_energy'][ok]
        print('  F err', (np.abs(Fa-Fb)/np.maximum(1,np.abs(Fa))).max())
    bad = np.nonzero(a['status']!=b['status'])[0][:3]
    for v in bad:
        print('  voxel', v, 'status', a['status'][v], b['status'][v], 'it', a['iterations'][v], b['iterations'][v], 'F', a['free_energy'][v], b['free_energy'][v])
        print('    o', a['mvn'][:,v]); print('    g', b['mvn'][:,v])
    return a, b
h,y = cases.exp_problem(2048-37, 100, 2, 0.02, seed=20260104, max_iterations=50); report('exp2', h, y)
h,y = cases.exp_problem(1500, 100, 2, 0.02, seed=7, convergence='pointzeroone', max_iterations=30); report('fchange', h, y)
h,y = cases.exp_problem(1500, 100, 2, 0.02, seed=7, convergence='trialmode', max_iterations=30); report('trialmode', h, y)
h,y = cases.exp_problem(700, 50, 1, 0.04, seed=11, max_iterations=12, need_f=True); report('F exp1', h, y)
h,y = cases.poly_problem(333, 24, 3, seed=21, max_iterations=15, masked_timepoints=(3, 7, 24), need_f=True); report('masked', h, y)
h,y = cases.exp_problem(500, 100, 2, 0.02, seed=13, max_iterations=5); f=oracle.run(h,y)
h2,_ = cases.exp_problem(500, 100, 2, 0.02, seed=13, max_iterations=5, init_mvn=f['mvn']); report('continue', h2, y)
