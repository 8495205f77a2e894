"""What the second-neighbour sweeps sum for every voxel in the first sweep (FVB_SLAB_DEBUG=256: contrib, contrib2, prior mean,
1/(8 nn - nn2), spatial precision, prior precision), slab form against data-flow form, on a tiny volume.

    python tools/measure/slab2_probe.py 3 1 1
"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 4:  # child: one run
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import numpy as np
    import test_spatial as ts
    from fabber_core_amd import hiplib, vbabi
    shape = tuple(int(a) for a in sys.argv[1:4])
    mask, coords = ts.masked_volume(shape, seed=31, keep=1.0)
    _, y = ts.smooth_exp_data(coords, 40, 0.04, seed=32)
    h = vbabi.build_config(vbabi.MODEL_EXP, coords.shape[1], 40, num_exps=1, dt=0.04, max_iterations=1, param_overrides={"amp1": dict(type="P")})
    hiplib.run_spatial_host(h, vbabi.SpatialHolder(coords), y)
    sys.exit(0)
out = {}
for mode, env in (("slab", {"FVB_SPATIAL_SLAB2": "1"}), ("flow", {"FVB_SPATIAL_SWEEP": "poll"})):
    e = dict(os.environ, FVB_SLAB_DEBUG="256", **env)
    p = subprocess.run([sys.executable, __file__] + sys.argv[1:4] + ["child"], env=e, capture_output=True, text=True)
    out[mode] = [l for l in p.stderr.splitlines() if l.startswith("[probe]")]
    if not out[mode]:
        print(mode, "no probe output;", p.stderr[-500:])
for a, b in zip(out["slab"], out["flow"]):
    fa, fb = a.split(), b.split()
    diff = [fa[i] for i in range(5, len(fa) - 1, 2) if fa[i + 1] != fb[i + 1]]
    print(("SAME " if not diff else "DIFF %s " % diff) + a[8:])
    if diff:
        print("       flow: " + b[8:])
