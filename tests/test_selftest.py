"""The reference's model-validation workflow (py/fabber.py:41-176: generate_test_data, self_test)
on the Python 3 client."""
import numpy as np
import pytest

from fabber_core_amd import fabber

EXP = {"model": "exp", "num-exps": 1, "dt": 0.04}


def test_generate_test_data_patches_follow_the_model():
    test = fabber.generate_test_data(EXP, {"amp1": [1.0, 0.5], "r1": [1.0, 0.8, 2.0]}, nt=20, patchsize=3)
    assert test["data"].shape == (6, 9, 3, 20) and test["data"] is test["clean"]
    t = np.arange(20) * 0.04
    for ia, amp in enumerate([1.0, 0.5]):
        for ir, r in enumerate([1.0, 0.8, 2.0]):
            block = test["clean"][ia * 3:(ia + 1) * 3, ir * 3:(ir + 1) * 3]
            assert np.allclose(block, amp * np.exp(-r * t), rtol=1e-6)
            assert (test["param_rois"]["amp1"][ia * 3:(ia + 1) * 3] == ia + 1).all()
            assert (test["param_rois"]["r1"][:, ir * 3:(ir + 1) * 3] == ir + 1).all()
    assert sorted(np.unique(test["patch_rois"])) == [1, 2, 3, 4, 5, 6]


def test_generate_test_data_noise_is_seeded_and_single_values_are_fixed():
    a = fabber.generate_test_data(EXP, {"amp1": 2.0, "r1": [1.0, 3.0]}, nt=10, patchsize=2, noise=0.1, seed=5)
    b = fabber.generate_test_data(EXP, {"amp1": 2.0, "r1": [1.0, 3.0]}, nt=10, patchsize=2, noise=0.1, seed=5)
    assert a["data"].shape == (4, 2, 2, 10) and list(a["param_rois"]) == ["r1"]
    assert np.array_equal(a["data"], b["data"])
    assert np.allclose(a["clean"][0, 0, 0, 0], 2.0)
    assert 0.05 < np.std(a["data"] - a["clean"]) < 0.2
    with pytest.raises(ValueError):
        fabber.generate_test_data(EXP, {"nosuch": [1, 2]})
    with pytest.raises(ValueError):
        fabber.generate_test_data({"model": "poly", "degree": 3}, {"c0": [1, 2], "c1": [1, 2], "c2": [1, 2], "c3": [1, 2]})


@pytest.mark.gpu
def test_self_test_recovers_the_test_values():
    report, log = fabber.self_test("exp", {"num-exps": 1, "dt": 0.04, "max-iterations": 20}, {"amp1": [1.0, 0.5], "r1": [1.0, 0.8]},
                                   nt=50, patchsize=8, noise=0.05, seed=11)
    for p in ("amp1", "r1"):
        for v, got in report[p].items():
            assert abs(got - v) < 0.05 * v, (p, v, got)
    (sd_in, sd_out), = report["noise"].items()
    assert sd_in == 0.05 and abs(sd_out - 0.05) < 0.005
    assert "exp" in log
