"""Evidence for the tolerance model of tests/parity.py: the REFERENCE ALGORITHM is numerically
chaotic on the bi-exponential fit (BASELINE config 3) and merely noisy on the others.

Two CPU builds of the same oracle source - with and without fused multiply-add contraction,
i.e. two equally valid compilations of the reference's arithmetic - are compared. Runs on CPU.
"""
import os
import sys

import numpy as np

import cases
import oracle
import parity

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def test_biexponential_fit_is_chaotic_between_two_cpu_builds():
    h, y = cases.exp_problem(400, 100, 2, 0.02, seed=20260103, max_iterations=50)
    a, b = oracle.run(h, y), oracle.run_fma(h, y)
    s = parity.population_stats(h, a, b)
    # a large share of voxels ends up more than the north star's 1e-4 apart ...
    assert s["frac_within_1e4"] < 0.95, s
    # ... although the typical voxel agrees to ~1e-9 and the populations are the same
    assert s["median_err"] < 1e-6, s
    assert np.allclose(s["phi_quantiles_a"], s["phi_quantiles_b"], rtol=0.05), s


def test_growth_of_rounding_differences_in_the_first_iterations():
    """The decay rates start at theta = 0, where the finite-difference step is at its 1e-10
    floor (fwdmodel_linear.cc:157-161): the very first Jacobian carries ~1e-6 relative rounding
    noise, so the two builds are already ~1e-8 apart after ONE iteration. The two exponentials
    also start with identical rates, J'J is numerically singular, and that difference grows by
    more than three orders of magnitude within four more iterations."""
    med = []
    for its in (1, 3, 5):
        h, y = cases.exp_problem(300, 100, 2, 0.02, seed=20260103, max_iterations=its)
        e_mean, _, _ = parity.voxel_errors(h, oracle.run(h, y), oracle.run_fma(h, y))
        med.append(np.median(e_mean))
    assert med[0] < 1e-6 and med[2] > 1e3 * med[0] and med[2] > 1e-4, med


def test_fp64_builds_against_the_binary128_ground_truth():
    """The same oracle source evaluated in binary128 (tests/golden/make_c3_truth.py) is what the
    ALGORITHM computes; both fp64 builds are measured against it on 4096 seeded voxels. The chaos is
    in the first iterations - the median error of the parameter means grows from 2e-7 after one
    iteration to 4e-2 after five - and then most voxels fall back onto the truth's fixed point: after
    50 iterations ~74 % of them are within the north star's 1e-4 (median 5e-10) and the rest sit on a
    different fixed point altogether (error ~10 posterior standard deviations). These are the numbers
    a GPU implementation is held to (tests/test_hip_parity.py::test_c3_error_against_the_ground_truth)."""
    truth = parity.load_c3_truth()
    import make_c3_truth as mt
    for run in (oracle.run, oracle.run_fma):
        h, y = mt.problem(truth["n_voxels"])
        s = parity.truth_stats(h, truth, run(h, y))
        assert 0.70 < s["within_1e4"] < 0.78 and s["median"] < 1e-9 and s["p90"] > 1.0 and s["failed"] < 3e-3, s
        med = {}
        for k, it in enumerate(truth["its"]):
            if it in (1, 5):
                hk, _ = mt.problem(truth["n_voxels"])
                hk.cfg.max_iterations = it
                med[it] = parity.truth_trace_stats(hk, truth["trace_means"][k], run(hk, y))["median"]
        assert med[1] < 1e-6 and med[5] > 1e-2, med


def test_well_conditioned_models_are_reproducible():
    for h, y in (cases.exp_problem(400, 50, 1, 0.04, seed=20260102, max_iterations=10),
                 cases.poly_problem(512, 10, 2, seed=20260101),
                 cases.linear_problem(300, 200, seed=20260104)):
        e_mean, e_cov, _ = parity.voxel_errors(h, oracle.run(h, y), oracle.run_fma(h, y))
        assert e_mean.max() < 1e-6 and e_cov.max() < 2e-4, (e_mean.max(), e_cov.max())


def _bad_fraction(env):
    """Fraction of voxels of the bi-exponential problem whose run ends with a non-finite
    prediction, in a fresh process (the oracle reads its experiment switches once)."""
    import json
    import os
    import subprocess
    import sys
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, cases, oracle\n"
            "h, y = cases.exp_problem(6000, 100, 2, 0.02, seed=20260103, max_iterations=50)\n"
            "print(json.dumps(float(np.mean(oracle.run(h, y)['status'] != 0))))\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, check=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_failing_voxels_follow_the_inversion_algorithm():
    """A few voxels of the bi-exponential fit reach a posterior precision whose condition number
    exceeds 1e16; what the "inverse" of such a matrix is depends on the algorithm. With the LU
    inverse (the oracle's restatement of NEWMAT .i()) some of them jump to a non-finite
    prediction and stop; with the symmetric sweep the kernels use (vb_math.h) none does. This is
    the documented difference between oracle and HIP path in the count of failed voxels
     - the GPU tests bound the kernels' count by the oracle's."""
    lu = _bad_fraction({})
    sweep = _bad_fraction({"ORACLE_SWEEP_INVERSE": "1"})
    assert 2e-4 < lu < 5e-3, lu
    assert sweep < lu / 4, (sweep, lu)
