"""Types P / p at C5's size. As the reference codes them (priors.cc:455, integer division) their prior mean reads no
neighbour's value: the split form has no ordered launch for them (prep + second sweep) - against the per-level launches,
which evaluate the reference's expression as it stands. The same problem, results compared.

    python tools/measure/spatial_second_neighbours.py [n=128] [iterations=10] [type=P]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases
from fabber_core_amd import hiplib, vbabi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
its = int(sys.argv[2]) if len(sys.argv) > 2 else 10
typ = sys.argv[3] if len(sys.argv) > 3 else "P"
res = {}
for label, env in (("split form (prep + second sweep)", {}), ("one launch per level", {"FVB_SPATIAL_PER_LEVEL": "1"})):
    os.environ.update(env)
    t = {}
    for k in (2, its):  # (two runs of different length: the difference is the iterations alone)
        h, coords, y, _ = cases.c5_problem((n, n, n), max_iterations=k, param_overrides={"amp1": dict(type=typ)})
        sp = vbabi.SpatialHolder(coords)
        hiplib.run_spatial_host(h, sp, y)
        t0 = time.perf_counter()
        r = hiplib.run_spatial_host(h, sp, y)
        t[k] = (time.perf_counter() - t0) * 1e3
    res[label] = r
    print("%s: %.1f ms per run of %d iterations, %.2f ms per iteration" % (label, t[its], its, (t[its] - t[2]) / (its - 2)), flush=True)
    for k in env:
        del os.environ[k]
ref = res["one launch per level"]
for label, r in res.items():
    if r is not ref:
        print(label + ":", "identical to the per-level sweep" if all(np.array_equal(r[k], ref[k], equal_nan=True) for k in ("mvn", "status")) else "DIFFERENT")
