"""Pin the oracle against the reference's own stored outputs (test/outdata_poly,
test/outdata_linear_vb; compared by the reference at 1e-3 in test/test_commandline.cc:10,69-93).

The input volume of those runs is missing from the reference snapshot, so the runs are replayed
from data with identical sufficient statistics (see tests/golden/make_golden.py).
"""
import numpy as np
import pytest

import golden_utils as gu
import oracle
from fabber_core_amd import vbabi

TOL = 1e-3  # the reference's own golden tolerance (test_commandline.cc:10)


@pytest.fixture(scope="module")
def ref():
    return gu.load_reference_outdata()


def _cases(ref):
    return {
        "poly": (gu.poly_design(106, 2), ref["poly/finalMVN"].astype(np.float64),
                 lambda V: vbabi.build_config(vbabi.MODEL_POLY, V, 106, degree=2)),
        "linear_vb": (ref["linear_design"], ref["linear_vb/finalMVN"].astype(np.float64),
                      lambda V: vbabi.build_config(vbabi.MODEL_LINEAR, V, 106, design=ref["linear_design"])),
    }


@pytest.mark.parametrize("name", ["poly", "linear_vb"])
def test_stored_posterior_satisfies_update_equations(ref, name):
    """eq (19) Lambda = L0 + phi J'J and eq (21) c = (T-1)/2 + c0 hold on the stored MVN:
    pins noisemodel_white.cc:292-305,263, the MVN inverse and the MVNDist::Save packing."""
    J, mvn, _ = _cases(ref)[name]
    P = J.shape[1]
    cov, means = gu.unpack(mvn, P + 1)
    G = J.T @ J
    for v in range(cov.shape[0]):
        phi = means[v, P]
        S = np.linalg.inv(1e-12 * np.eye(P) + phi * G)
        assert np.max(np.abs(S - cov[v, :P, :P]) / np.abs(S)) < 1e-5
        b = cov[v, P, P] / phi
        c = phi / b
        assert abs(c - (105 * 0.5 + 1e-6)) < 1e-4
        assert np.all(cov[v, P, :P] == 0)


@pytest.mark.parametrize("name", ["poly", "linear_vb"])
def test_oracle_reproduces_stored_fixed_point(ref, name):
    J, mvn, mk = _cases(ref)[name]
    P = J.shape[1]
    cov, means = gu.unpack(mvn, P + 1)
    y = gu.data_with_same_sufficient_statistics(J, cov, means)
    h = mk(y.shape[1])
    res = oracle.run(h, y)
    assert np.all(res["status"] == 0) and np.all(res["iterations"] == 10)
    got_cov, got_means = gu.unpack(res["mvn"], P + 1)
    sd = np.sqrt(np.einsum("vii->vi", cov))
    # means within 1e-3 posterior standard deviations and 1e-3 relative (float32 storage)
    assert np.max(np.abs(got_means - means) / np.maximum(sd, np.abs(means) * 1.0)) < TOL
    scale = sd[:, :, None] * sd[:, None, :]
    assert np.max(np.abs(got_cov - cov) / scale) < TOL
    # tight check on the well-conditioned numbers: noise mean/variance and variances
    assert np.max(np.abs(got_means[:, P] / means[:, P] - 1)) < 1e-5
    assert np.max(np.abs(np.einsum("vii->vi", got_cov) / np.einsum("vii->vi", cov) - 1)) < 1e-5


def test_postproc_matches_stored_images(ref):
    """InferenceTechnique::SaveResults / Vb::SaveResults on the stored finalMVN give the stored
    mean_/std_/zstat_/noise_ images (inference.cc:139-155, inference_vb.cc:981-989)."""
    mvn = ref["poly/finalMVN"].astype(np.float64)
    V = mvn.shape[1]
    h = vbabi.build_config(vbabi.MODEL_POLY, V, 106, degree=2)
    pp = oracle.postproc(h, np.zeros((106, V), dtype=np.float32), mvn,
                         want=("mean", "std", "zstat", "noise_mean", "noise_std"))
    for p in range(3):
        assert np.allclose(pp["mean"][p], ref["poly/mean_c%d" % p][0], rtol=1e-6, atol=0)
        assert np.allclose(pp["std"][p], ref["poly/std_c%d" % p][0], rtol=1e-5, atol=0)
        assert np.allclose(pp["zstat"][p], ref["poly/zstat_c%d" % p][0], rtol=1e-4, atol=1e-4)
    assert np.allclose(pp["noise_mean"][0], ref["poly/noise_means"][0], rtol=1e-6)
    assert np.allclose(pp["noise_std"][0], ref["poly/noise_stdevs"][0], rtol=1e-5)
