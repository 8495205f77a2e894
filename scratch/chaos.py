import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import ctypes as C, numpy as np, time
import cases, oracle
from fabber_core_amd import vbabi
L2 = C.CDLL('/tmp/liboracle_fma.so')
L2.oracle_vb_run.restype = C.c_int32
L2.oracle_vb_run.argtypes = [C.POINTER(vbabi.FvbConfig), C.c_void_p, C.POINTER(vbabi.FvbOutputs), C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
def run2(h, y):
    y = oracle.prepare_data(h, y); arrs, out = oracle.alloc_outputs(h)
    L2.oracle_vb_run(C.byref(h.cfg), y.ctypes.data, C.byref(out), 0, h.cfg.n_voxels, 0, None)
    return arrs
V=3000
for its in (10, 50):
  for ne, T, dt in ((1,50,0.04),(2,100,0.02)):
    h,y = cases.exp_problem(V, T, ne, dt, seed=20260103, max_iterations=its)
    t0=time.time(); a = oracle.run(h,y); t1=time.time(); b = run2(h,y)
    n = 2*ne+1
    ca, ma = oracle.unpack_mvn(a['mvn'], n); cb, mb = oracle.unpack_mvn(b['mvn'], n)
    rel = (np.abs(ma-mb)/np.maximum(np.abs(ma),1e-12))[:, :2*ne].max(axis=1)
    print('exp%d its=%d  oracle %.1f vox/s' % (ne, its, V/(t1-t0)), 'bad', (a['status']!=0).sum(), (b['status']!=0).sum(),
          'frac rel>1e-4: %.4f  >1e-6: %.4f  >1e-8: %.4f  median %.2e' % ((rel>1e-4).mean(), (rel>1e-6).mean(), (rel>1e-8).mean(), np.median(rel)))
