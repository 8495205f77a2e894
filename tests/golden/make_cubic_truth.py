#!/usr/bin/env python3
"""Ground truth in IEEE binary128 for the cubic-polynomial cases of tests/cases.py:cubic_cases() - the voxelwise loop
with ARD priors, masked timepoints, prior-noise-stddev and locked-noise-stdev on a design whose columns span four orders
of magnitude, where two fp64 builds of the oracle are 3e-6 - 3e-5 apart on single voxels. The oracle source built with
every internal variable in binary128 (oracle/Makefile: liboracle_quad.so) gives what the ALGORITHM computes; the fp64
implementations - both CPU builds, the HIP kernels - are then each held to their distance from it
(tests/test_hip_parity.py::test_cubic_cases_against_the_ground_truth).

    python tests/golden/make_cubic_truth.py      -> tests/golden/cubic_truth_binary128.npz
      <case>/mvn [rows][V], <case>/free_energy [V], <case>/status [V], <case>/iterations [V], <case>/data_sha256
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    import cases
    import oracle
    out = {}
    for name, (h, y) in cases.cubic_cases().items():
        r = oracle.run_quad(h, y)
        key = name.replace(" ", "_")
        for k in ("mvn", "free_energy", "status", "iterations"):
            out[key + "/" + k] = r[k]
        out[key + "/data_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest())
        print(name, "done: failed voxels", int(np.count_nonzero(r["status"])), file=sys.stderr, flush=True)
    path = os.path.join(HERE, "cubic_truth_binary128.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
