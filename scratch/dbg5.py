import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import cases, oracle, parity
V=900
for its in (2,4,8,12,30):
    h, y = cases.poly_problem(V, 20, 3, seed=6, max_iterations=its, need_f=True, param_overrides={"c1": dict(type="A"), "c2": dict(mean=1.0, prec=0.5)})
    a, b = oracle.run(h,y), oracle.run_fma(h,y)
    e_mean, e_cov, rel = parity.voxel_errors(h, a, b)
    print(its, 'floor means q50 %.1e q99 %.1e max %.1e  cov max %.1e' % (np.median(e_mean), np.quantile(e_mean,.99), e_mean.max(), e_cov.max()))
