"""The slab form of the second-neighbour sweep (FVB_SPATIAL_SLAB2=1, vb_spatial_slab2_sweep_kernel; not what runs: DESIGN 3.4)
against the data-flow form on small line / plane / box volumes: which voxels differ after one iteration."""
import os, sys
ROOT="/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_spatial as ts
from fabber_core_amd import hiplib, vbabi
def run(shape, env={}):
    mask, coords = ts.masked_volume(shape, seed=31, keep=1.0)
    _, y = ts.smooth_exp_data(coords, 40, 0.04, seed=32)
    h = vbabi.build_config(vbabi.MODEL_EXP, coords.shape[1], 40, num_exps=1, dt=0.04, max_iterations=1, param_overrides={"amp1": dict(type="P")})
    sp = vbabi.SpatialHolder(coords)
    os.environ["FVB_SPATIAL_SWEEP"]="poll"
    ref = hiplib.run_spatial_host(h, sp, y)
    del os.environ["FVB_SPATIAL_SWEEP"]
    os.environ.update(env)
    os.environ["FVB_SPATIAL_SLAB2"] = "1"
    r = hiplib.run_spatial_host(h, sp, y)
    del os.environ["FVB_SPATIAL_SLAB2"]
    for k in env: del os.environ[k]
    bad = np.flatnonzero(np.any(ref["mvn"] != r["mvn"], axis=0))
    print(shape, env, len(bad), "of", coords.shape[1], "differ:", [tuple(int(c) for c in coords[:, v]) for v in bad[:10]], flush=True)
for s in ((3,1,1),(5,1,1),(9,1,1),(17,1,1),(20,1,1)):
    run(s)
run((20,1,1), {"FVB_SPATIAL_SLAB_WIDTH":"512"})
for s in ((1,3,1),(1,5,1),(1,9,1)):
    run(s)
run((1,9,1), {"FVB_SPATIAL_SLAB_WIDTH":"512"})
run((2,2,1)); run((3,3,1)); run((2,1,2)); run((1,2,2))
