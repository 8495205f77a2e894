import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import cases, oracle, hipengine, parity
from fabber_core_amd import hiplib, vbabi
np.set_printoptions(precision=3, linewidth=220)
V=1037
for k in (0,1,2,3,5,8,13,21,34,49):
    h, y = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=max(k, 1))
    st = oracle.run(h, y)
    state = st['mvn'] if k>0 else None
    h1, _ = cases.exp_problem(V, 100, 2, 0.02, seed=20260103, max_iterations=1, init_mvn=state, need_f=True)
    a = oracle.run(h1, y); b = hipengine.run(h1, y)
    ok = np.isfinite(a['mvn']).all(axis=0) & (a['status']==0) & (b['status']==0)
    e_mean, e_cov, _ = parity.voxel_errors(h1, a, b, ok)
    n=5; off=15
    th_in = np.abs(state[off:off+4][:,ok]).max(axis=0) if k>0 else np.zeros(ok.sum())
    th_out = np.abs(a['mvn'][off:off+4][:,ok]).max(axis=0)
    wild = np.maximum(th_in, th_out) > 6
    ca, ma = oracle.unpack_mvn(a['mvn'][:,ok], n); cb, mb = oracle.unpack_mvn(b['mvn'][:,ok], n)
    sd = np.sqrt(np.abs(np.einsum('vii->vi', ca)))
    em = np.abs(ma-mb)/np.maximum(np.abs(ma), sd)
    print('k=%2d ok %.3f statusmis %d | e_mean q50 %.1e q90 %.1e q99 %.1e max %.1e | wild frac %.3f | tame max %.1e | per-col q99' % (k, ok.mean(), (a['status']!=b['status']).sum(), *np.quantile(e_mean,[.5,.9,.99]), e_mean.max(), wild.mean(), e_mean[~wild].max() if (~wild).any() else 0), np.quantile(em,0.99,axis=0))
