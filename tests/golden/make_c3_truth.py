#!/usr/bin/env python3
"""Ground truth for the bi-exponential fit (BASELINE configs[2] model) in IEEE binary128.

The reference's algorithm is numerically chaotic on this problem (tests/test_reference_chaos.py):
two fp64 builds of the same statements end up more than 1e-4 apart on a quarter of the voxels.
To say how far any fp64 implementation - the CPU oracle, its FMA build, the HIP kernels - is from
what the ALGORITHM computes, the oracle source is built a third time with every internal variable
in binary128 (oracle/Makefile: liboracle_quad.so, libquadmath) and run on a fixed seeded sample.
Rounding errors of ~1e-34 stay below 1e-15 after the ~1e11-fold amplification of the first
iterations, so the result is exact at fp64 resolution.

binary128 runs at ~8 voxels/s per core, so the result is committed as a fixture:
    tests/golden/c3_truth_binary128.npz
      mvn [21][V], status [V], iterations [V]   final posterior of the 50-iteration run
      free_energy [V]                           F of that posterior (noisemodel_white.cc:365-454 + the priors' term)
      its [K], trace_means [K][4][V]            posterior means after its[k] iterations
      data_sha256                               of the float32 series cases.exp_problem generates

    python tests/golden/make_c3_truth.py [--voxels 4096] [--jobs 8]
"""
import argparse
import concurrent.futures
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

SEED, T, DT, MAX_ITS = 20260103, 100, 0.02, 50
ITS = [1, 2, 3, 5, 10, 20, 35, 50]


def problem(n_voxels, **kw):
    import cases
    return cases.exp_problem(n_voxels, T, 2, DT, seed=SEED, max_iterations=MAX_ITS, **kw)


def _block(args):
    n_voxels, v0, v1 = args
    import oracle
    # (with the counting detector F does not steer the loop: the posterior is the same with and without it)
    h, y = problem(n_voxels, need_f=True)
    r = oracle.run_quad(h, y, v_begin=v0, v_end=v1, trace_rows=MAX_ITS)
    return v0, v1, {k: (r[k][..., v0:v1]) for k in ("mvn", "status", "iterations", "trace_means", "free_energy")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=4096)
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 4)
    a = ap.parse_args()
    V = a.voxels
    h, y = problem(V)
    step = 32
    work = [(V, v0, min(V, v0 + step)) for v0 in range(0, V, step)]
    mvn = np.full((21, V), np.nan)
    status = np.full(V, -1, dtype=np.int32)
    iters = np.full(V, -1, dtype=np.int32)
    trace = np.full((len(ITS), 4, V), np.nan)
    free_energy = np.full(V, np.nan)
    done = 0
    with concurrent.futures.ProcessPoolExecutor(max_workers=a.jobs) as ex:
        for v0, v1, r in ex.map(_block, work):
            mvn[:, v0:v1] = r["mvn"]
            status[v0:v1] = r["status"]
            iters[v0:v1] = r["iterations"]
            free_energy[v0:v1] = r["free_energy"]
            for k, it in enumerate(ITS):
                trace[k, :, v0:v1] = r["trace_means"][it - 1]
            done += v1 - v0
            print("\r%d / %d voxels" % (done, V), end="", file=sys.stderr, flush=True)
    print(file=sys.stderr)
    out = os.path.join(HERE, "c3_truth_binary128.npz")
    np.savez_compressed(out, mvn=mvn, status=status, iterations=iters, its=np.array(ITS), trace_means=trace, free_energy=free_energy,
                        data_sha256=np.array(hashlib.sha256(y.tobytes()).hexdigest()), n_voxels=np.array(V))
    print("wrote", out, os.path.getsize(out), "bytes; failed voxels:", int(np.count_nonzero(status)))


if __name__ == "__main__":
    main()
