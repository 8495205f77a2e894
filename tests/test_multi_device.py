"""Sharding inside the C++ engine (SURVEY 8e "voxelwise VB: independent units"): fabber_vb_run_host_multi
cuts the voxel list into contiguous blocks, one host thread + stream per block, block i on device devices[i],
nothing exchanged between blocks; through the reference's C API it is the option devices=all | 0,1,...
The GPU box of the tests has ONE device: `all` must reproduce device 0 bit for bit, and a device listed
several times rehearses the N-block code path (blocks on separate streams of the same GPU)."""
import numpy as np
import pytest

import cases
from fabber_core_amd import fabber, hiplib

pytestmark = pytest.mark.gpu


def same(a, b):
    for k in ("mvn", "free_energy", "status", "iterations", "f_history_len"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("need_f", [False, True])
def test_all_devices_equals_device_zero(need_f):
    h, y = cases.exp_problem(20000 + 17, 100, 2, 0.02, seed=11, max_iterations=12, need_f=need_f)
    one = hiplib.run_host(h, y)
    multi = hiplib.run_host(h, y, devices="all")
    same(one, multi)
    s_f, s_it, bad = multi["summary"]
    ok = one["status"] == 0
    assert s_it == int(one["iterations"].sum()) and bad == int(np.count_nonzero(~ok))
    if need_f:
        # added up in voxel order on the host (np.sum adds pairwise: with a voxel whose F is 1e48 the two orders differ)
        assert s_f == float(np.cumsum(one["free_energy"][ok])[-1])


@pytest.mark.parametrize("blocks", [2, 3, 5])
def test_blocks_on_streams_of_one_device(blocks):
    """contiguous blocks cut on wavefront boundaries, ragged tail, convergence detector with save / revert,
    image prior and masked timepoints (the strided kernel) - every per-voxel input has to be cut too"""
    V = 9000 + 33
    rng = np.random.default_rng(5)
    img = rng.normal(0.5, 0.1, V)
    h, y = cases.poly_problem(V, 24, 2, seed=4, max_iterations=12, need_f=True, convergence="trialmode",
                              param_overrides={"c1": dict(type="I", prec=4.0)}, image_priors={"c1": img},
                              masked_timepoints=(3, 7), f_history_rows=14)
    hiplib.set_variant("lane")
    try:
        one = hiplib.run_host(h, y)
        multi = hiplib.run_host(h, y, devices=[0] * blocks)
    finally:
        hiplib.set_variant("auto")
    same(one, multi)
    assert np.array_equal(one["f_history"], multi["f_history"], equal_nan=True)


def test_small_problem_keeps_the_kernel_of_the_whole():
    """3000 voxels take the wave-per-voxel kernel; its blocks must not switch to another one"""
    h, y = cases.exp_problem(3000, 50, 1, 0.04, seed=3, max_iterations=10)
    assert hiplib.kernel_name(h) == "wave"
    same(hiplib.run_host(h, y), hiplib.run_host(h, y, devices=[0, 0, 0]))


def test_continue_from_mvn_is_cut_with_the_blocks():
    h, y = cases.exp_problem(8192 + 5, 50, 1, 0.04, seed=13, max_iterations=4)
    first = hiplib.run_host(h, y)
    h2, _ = cases.exp_problem(8192 + 5, 50, 1, 0.04, seed=13, max_iterations=4, init_mvn=first["mvn"])
    same(hiplib.run_host(h2, y), hiplib.run_host(h2, y, devices=[0, 0]))


def test_bad_device_index_is_an_error():
    h, y = cases.poly_problem(128, 10, 1, seed=1)
    with pytest.raises(hiplib.HipEngineError, match="out of range"):
        hiplib.run_host(h, y, devices=[0, 99])


def test_devices_option_through_the_c_api():
    """the reference's fabber_dorun with devices=all / devices=0,0 gives the images of device=0"""
    rng = np.random.default_rng(2)
    shape = (24, 20, 12)
    t = np.arange(1, 11, dtype=np.float64)
    coef = rng.uniform(-2, 2, (3,) + shape)
    data = (coef[0][..., None] + coef[1][..., None] * t + coef[2][..., None] * t * t + rng.normal(0, 0.05, shape + (10,))).astype(np.float32)
    opts = {"model": "poly", "degree": 2, "noise": "white", "method": "vb", "max-iterations": 6, "save-mean": True, "save-mvn": True}
    ref = fabber.run(data, opts)
    for devices in ("all", "0,0"):
        got = fabber.run(data, dict(opts, devices=devices))
        assert set(got) == set(ref)
        for k in ref:
            if k != "log":  # (the log names the option and the time of day)
                assert np.array_equal(ref[k], got[k]), (devices, k)
        assert "devices=" + devices in got["log"]
    with pytest.raises(Exception, match="devices"):
        fabber.run(data, dict(opts, devices="0,,1"))


def test_several_noise_precisions_are_cut_with_the_blocks():
    """noise-pattern on the lane kernel with per-precision moments, the voxel list cut into blocks"""
    rng = np.random.default_rng(12)
    h, y = cases.poly_problem(5000, 24, 2, seed=13, noise_pattern="123", max_iterations=6, need_f=True)
    y = y.astype(np.float64) + rng.normal(0, 0.05, y.shape) * (1 + np.arange(24)[:, None] % 3)
    assert hiplib.kernel_name(h) == "lane_phis<poly,3,4>"
    same(hiplib.run_host(h, y), hiplib.run_host(h, y, devices=[0, 0, 0]))


def test_cached_device_memory_can_be_given_back():
    """the engine's work buffers come from the device's memory pool, which keeps them between runs;
    fabber_vb_release_cached_memory returns them - and the next run allocates again"""
    h, y = cases.exp_problem(8192, 50, 1, 0.04, seed=3, max_iterations=5)
    first = hiplib.run_host(h, y)
    lib = hiplib.lib()
    lib.fabber_vb_release_cached_memory.restype = None
    lib.fabber_vb_release_cached_memory()
    same(first, hiplib.run_host(h, y))


@pytest.mark.parametrize("case", ["plain", "detector + image prior + continue"])
def test_the_host_entry_point_pipelines_over_blocks(case, monkeypatch):
    """fabber_vb_run_host on a large problem: blocks of voxels go up, are fitted and come down on three streams, the
    uploads and downloads of neighbouring blocks hidden behind the arithmetic (vb_api.hip, run_host_pipelined). The
    result is the one-block run's bit for bit - here with small blocks (many of them, a ragged last one) and at the
    default block size."""
    V = 50000 + 21
    if case == "plain":
        h, y = cases.exp_problem(V, 50, 1, 0.04, seed=3, max_iterations=6)
    else:
        rng = np.random.default_rng(6)
        img = rng.normal(0.5, 0.1, V)
        h0, y = cases.poly_problem(V, 16, 2, seed=4, max_iterations=3)
        first = hiplib.run_host(h0, y)
        h, y = cases.poly_problem(V, 16, 2, seed=4, max_iterations=8, need_f=True, convergence="trialmode", f_history_rows=10,
                                  param_overrides={"c1": dict(type="I", prec=4.0)}, image_priors={"c1": img}, init_mvn=first["mvn"])
    monkeypatch.setenv("FVB_HOST_BLOCK_VOXELS", "0")
    one = hiplib.run_host(h, y)
    monkeypatch.setenv("FVB_HOST_BLOCK_VOXELS", "4096")
    piped = hiplib.run_host(h, y)
    same(one, piped)
    if "f_history" in one:
        assert np.array_equal(one["f_history"], piped["f_history"], equal_nan=True)
    monkeypatch.delenv("FVB_HOST_BLOCK_VOXELS")
    if case == "plain":
        hb, yb = cases.exp_problem(2 * 262144 + 4097, 20, 1, 0.1, seed=5, max_iterations=3)
        monkeypatch.setenv("FVB_HOST_BLOCK_VOXELS", "0")
        one = hiplib.run_host(hb, yb)
        monkeypatch.delenv("FVB_HOST_BLOCK_VOXELS")
        same(one, hiplib.run_host(hb, yb))
