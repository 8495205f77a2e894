import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import cases, oracle, hipengine
from fabber_core_amd import hiplib, vbabi
np.set_printoptions(precision=6, linewidth=200)
h, y = cases.poly_problem(512, 10, 2, seed=20260101)
a = oracle.run(h, y, trace_rows=10); b = hipengine.run(h, y)
rel = np.abs(a['mvn']-b['mvn'])/np.maximum(np.abs(a['mvn']),1e-300)
print('rel err per row (max over voxels):'); print(rel.max(axis=1))
v = np.argmax(rel.max(axis=0)); print('worst voxel', v); print(a['mvn'][:,v]); print(b['mvn'][:,v])
for its in (1,2,3,5,10):
    h2, _ = cases.poly_problem(512, 10, 2, seed=20260101, max_iterations=its)
    a2 = oracle.run(h2, y); b2 = hipengine.run(h2, y)
    rel2 = np.abs(a2['mvn']-b2['mvn'])/np.maximum(np.abs(a2['mvn']),1e-300)
    print(its, rel2.max(axis=1))
