import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import cases, oracle, hipengine
from fabber_core_amd import hiplib, vbabi
np.set_printoptions(precision=6, linewidth=220)
V=256
for its in (1,2,3,4,5,6,8,10,15,20,30,50):
    h,y = cases.exp_problem(V, 100, 2, 0.02, seed=20260104, max_iterations=its)
    a = oracle.run(h, y); b = hipengine.run(h, y)
    n=5
    ca, ma = oracle.unpack_mvn(a['mvn'], n); cb, mb = oracle.unpack_mvn(b['mvn'], n)
    sd = np.sqrt(np.abs(np.einsum('vii->vi', ca)))
    em = np.abs(ma-mb)/np.maximum(np.abs(ma), sd)
    print(its, 'status', (a['status']!=0).sum(), (b['status']!=0).sum(), 'err means: median', np.median(em,axis=0), 'max', em.max(axis=0))
    if its in (1,2,3): print('   v0 oracle', ma[0], 'gpu', mb[0])
