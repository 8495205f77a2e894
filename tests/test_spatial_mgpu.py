"""Spatial VB over several GPUs (fabber_core_amd/spatial_mgpu.py): slab decomposition, the pipeline schedule
of the first sweep and the message pattern with gloo ranks on CPU; on the GPU, two and three processes
(sharing the one card of the test box) must reproduce the single-process run BIT FOR BIT: the slabs sweep
the same global levels as a pipeline, which is the reference's Gauss-Seidel order voxel for voxel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from fabber_core_amd import hiplib, spatial_mgpu, vbabi  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def masked_coords(shape, seed, keep=0.85):
    rng = np.random.default_rng(seed)
    return vbabi.grid_coords(shape, rng.random(shape) < keep)


# ---------------------------------------------------------------------------------------------
# CPU
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world,halo", [(1, 1), (2, 1), (3, 1), (4, 2)])
def test_slab_plan_covers_the_volume_with_whole_planes(world, halo):
    coords = masked_coords((7, 6, 16), seed=world)
    V = coords.shape[1]
    plan = spatial_mgpu.slab_plan(coords, world, halo)
    z = coords[2]
    assert plan[0][1] == 0 and plan[-1][2] == V
    for r, (g0, b, e, g1) in enumerate(plan):
        assert g0 <= b < e <= g1
        if r:
            assert plan[r - 1][2] == b                       # contiguous, disjoint
            assert z[b] != z[b - 1]                           # cut on a plane boundary
            assert set(z[g0:b]) == set(range(z[b] - halo, z[b]))
        else:
            assert g0 == b
        if r < world - 1:
            assert set(z[e:g1]) == set(range(z[e - 1] + 1, z[e - 1] + 1 + halo))
        else:
            assert g1 == e
        assert abs((e - b) - V / world) < 2 * 7 * 6         # balanced to within a plane or two


def test_slab_plan_rejects_what_it_cannot_cut():
    coords = vbabi.grid_coords((4, 4, 3))
    with pytest.raises(ValueError, match="too few z-planes"):
        spatial_mgpu.slab_plan(coords, 4, 1)
    with pytest.raises(ValueError, match="thinner than"):
        spatial_mgpu.slab_plan(vbabi.grid_coords((4, 4, 4)), 4, 2)
    with pytest.raises(ValueError, match="z slowest"):
        spatial_mgpu.slab_plan(coords[:, ::-1], 2, 1)


def test_local_holder_slices_per_voxel_arrays():
    V = 50
    img = np.arange(V, dtype=np.float64)
    mvn = np.arange(10 * V, dtype=np.float64).reshape(10, V)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 20, num_exps=1, dt=0.1, param_overrides={"r1": dict(type="I", prec=2.0)},
                           image_priors={"r1": img}, init_mvn=mvn)
    loc = spatial_mgpu.local_holder(h, 10, 30)
    assert loc.cfg.n_voxels == 20 and h.cfg.n_voxels == V
    assert np.array_equal(loc.keep["init_mvn"], mvn[:, 10:30])
    assert np.array_equal(loc.keep["image_1"], img[10:30])
    assert spatial_mgpu.halo_planes(h) == 1
    hp = vbabi.build_config(vbabi.MODEL_EXP, V, 20, num_exps=1, dt=0.1, param_overrides={"amp1": dict(type="P")})
    assert spatial_mgpu.halo_planes(hp) == 2


class FakeRun:
    """The C run's copy_means on a numpy image: means[p][v] = 1000 rank + 10 p + global v."""

    def __init__(self, rank, g0, g1, b, e, P):
        self.g0 = g0
        v = np.arange(g0, g1)
        self.means = np.stack([1000.0 * rank + 10 * p + v for p in range(P)])
        self.status = np.where((v >= b) & (v < e), rank, -1).astype(np.int32)  # ghosts start "unknown"

    def get(self, v0, n):
        return self.means[:, v0:v0 + n].copy(), self.status[v0:v0 + n].copy()

    def put(self, v0, means, status):
        self.means[:, v0:v0 + status.shape[0]] = means
        self.status[v0:v0 + status.shape[0]] = status


def _exchange_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coords = masked_coords((5, 4, 12), seed=5)
    P = 2
    plan = spatial_mgpu.slab_plan(coords, world, 2)
    g0, b, e, g1 = plan[rank]
    run = FakeRun(rank, g0, g1, b, e, P)
    spatial_mgpu._exchange(run, plan, rank, world, P, "cpu")
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), means=run.means, status=run.status, plan=np.array(plan))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_between_gloo_ranks(tmp_path):
    world = 3
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    plan = res[0]["plan"]
    owner = np.zeros(plan[-1][2], dtype=int)
    for r, (g0, b, e, g1) in enumerate(plan):
        owner[b:e] = r
    for r, (g0, b, e, g1) in enumerate(plan):
        v = np.arange(g0, g1)
        assert np.array_equal(res[r]["status"], owner[v])              # every ghost now carries its owner's status
        for p in range(2):
            assert np.array_equal(res[r]["means"][p], 1000.0 * owner[v] + 10 * p + v)


# ---------------------------------------------------------------------------------------------
# GPU: two processes on the one card, gloo for the collectives
# ---------------------------------------------------------------------------------------------
def spatial_problem(typ, need_f=False):
    coords = masked_coords((10, 9, 12), seed=21, keep=0.9)
    V = coords.shape[1]
    rng = np.random.default_rng(22)
    T = 50
    t = np.arange(T) * 0.04
    amp = 1.0 + 0.3 * np.sin(coords[0] / 3.0) * np.cos(coords[1] / 4.0) + 0.2 * np.sin(coords[2] / 2.0)
    y = amp[None, :] * np.exp(-t[:, None]) + rng.normal(0, 0.1, (T, V))
    h = vbabi.build_config(vbabi.MODEL_EXP, V, T, num_exps=1, dt=0.04, max_iterations=8, need_f=need_f,
                           param_overrides={"amp1": dict(type=typ)})
    return h, vbabi.SpatialHolder(coords), y


def _gpu_worker(rank, world, port, typ, out_dir):
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h, sp, y = spatial_problem(typ)
    res = spatial_mgpu.run_spatial_sharded(h, sp, y, device="cuda:0")
    np.savez(os.path.join(out_dir, "g%d.npz" % rank), **{k: v for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_one_rank_is_the_single_device_run():
    h, sp, y = spatial_problem("M", need_f=True)
    one = spatial_mgpu.run_spatial_sharded(h, sp, y)
    ref = hiplib.run_spatial_host(h, sp, y)
    assert one["begin"] == 0 and one["end"] == h.cfg.n_voxels
    assert np.array_equal(one["mvn"], ref["mvn"]) and np.array_equal(one["free_energy"], ref["free_energy"])


def test_pipeline_schedule_respects_the_sweep_order():
    """slab r sweeps chunk c at tick c + r: whatever a voxel needs from the slab below (a lower level) was swept
    there at an earlier tick, every slab sweeps every chunk once, in order"""
    for world, lmin, lmax, C in ((2, 0, 40, 16), (3, 5, 100, 7), (8, 0, 381, 16), (4, 0, 3, 16)):
        ticks = spatial_mgpu.pipeline_schedule(lmin, lmax, C, world)
        done = {r: [] for r in range(world)}
        for t, work in ticks:
            for r, (lo, hi) in work.items():
                if r > 0:  # the slab below has finished every level below hi - 1 (lower neighbours have lower levels)
                    assert done[r - 1] and done[r - 1][-1][1] >= hi, (world, t, r)
                done[r].append((lo, hi))
        for r in range(world):
            assert done[r][0][0] == lmin and done[r][-1][1] > lmax
            assert all(done[r][i][1] == done[r][i + 1][0] for i in range(len(done[r]) - 1))
        assert len(ticks) == len(done[0]) + world - 1


def _pipeline_worker(rank, world, port, out_dir):
    """the hand-overs of one first sweep with fake runs: every message must find its partner (no deadlock) and
    every ghost-below plane must end up with the value its owner had AFTER the owner's last sweep step"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coords = masked_coords((5, 4, 12), seed=5)
    P = 2
    plan = spatial_mgpu.slab_plan(coords, world, 1)
    g0, b, e, g1 = plan[rank]
    run = FakeRun(rank, g0, g1, b, e, P)
    level = coords[0].astype(np.int64) + coords[1] + coords[2]
    for t, work in spatial_mgpu.pipeline_schedule(int(level.min()), int(level.max()), 4, world):
        if rank in work:  # "sweep": stamp the owned voxels of these levels with the tick
            lo, hi = work[rank]
            v = np.arange(g0, g1)
            hit = (v >= b) & (v < e) & (level[g0:g1] >= lo) & (level[g0:g1] < hi)
            run.means[:, hit] = 100.0 * t + rank
        spatial_mgpu._exchange_up(run, plan, rank, world, P, "cpu", rank in work and rank < world - 1, (rank - 1) in work)
    np.savez(os.path.join(out_dir, "p%d.npz" % rank), means=run.means, plan=np.array(plan))
    dist.barrier()
    dist.destroy_process_group()


def test_pipeline_hand_overs_between_gloo_ranks(tmp_path):
    world = 3
    mp.spawn(_pipeline_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "p%d.npz" % r)) for r in range(world)]
    plan = res[0]["plan"]
    for r in range(1, world):
        g0, b, e, g1 = plan[r]
        pg0 = plan[r - 1][0]
        # rank r's ghosts below = rank r-1's owned top plane, as rank r-1 left it
        assert np.array_equal(res[r]["means"][:, :b - g0], res[r - 1]["means"][:, g0 - pg0:b - pg0])


@pytest.mark.gpu
@pytest.mark.parametrize("typ,world", [("M", 2), ("P", 2), ("M", 3)])
def test_slabs_against_the_single_device_run(tmp_path, typ, world):
    """Two or three ranks: posterior, status and iteration counts of every voxel identical to the single-device
    run (= the reference's sweep order), first- and second-neighbour priors."""
    sys.path.insert(0, HERE)
    mp.spawn(_gpu_worker, args=(world, _free_port(), typ, str(tmp_path)), nprocs=world, join=True)
    h, sp, y = spatial_problem(typ)
    ref = hiplib.run_spatial_host(h, sp, y)
    parts = [np.load(os.path.join(str(tmp_path), "g%d.npz" % r)) for r in range(world)]
    assert [int(p["begin"]) for p in parts][0] == 0 and int(parts[-1]["end"]) == h.cfg.n_voxels
    got = dict(mvn=np.concatenate([p["mvn"] for p in parts], axis=1), status=np.concatenate([p["status"] for p in parts]),
               iterations=np.concatenate([p["iterations"] for p in parts]))
    assert np.all(got["status"] == 0) and np.all(got["iterations"] == 8)
    for k in got:
        assert np.array_equal(got[k], ref[k]), (typ, world, k)


def _gpu_worker_f(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h, sp, y = ard_problem()
    res = spatial_mgpu.run_spatial_sharded(h, sp, y, device="cuda:0", chunk_levels=5)
    np.savez(os.path.join(out_dir, "f%d.npz" % rank), **{k: v for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


def ard_problem():
    """free energy on, an ARD prior (its F term is the LAST voxel's for every voxel) next to the spatial one,
    a voxel that fails during the first sweep of the second slab"""
    coords = masked_coords((8, 7, 9), seed=23, keep=0.9)
    V = coords.shape[1]
    rng = np.random.default_rng(24)
    t = np.arange(1, 21.0)
    c0 = 2.0 + np.sin(coords[0] / 2.0) + 0.3 * coords[2]
    y = c0[None, :] + 0.3 * t[:, None] + rng.normal(0, 0.2, (20, V))
    y[5, (2 * V) // 3] = np.nan
    h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=1, max_iterations=5, need_f=True,
                           param_overrides={"c0": dict(type="M"), "c1": dict(type="A")})
    return h, vbabi.SpatialHolder(coords, update_first_iter=True), y


@pytest.mark.gpu
def test_slabs_with_free_energy_ard_and_a_failing_voxel(tmp_path):
    world = 2
    mp.spawn(_gpu_worker_f, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    h, sp, y = ard_problem()
    ref = hiplib.run_spatial_host(h, sp, y)
    parts = [np.load(os.path.join(str(tmp_path), "f%d.npz" % r)) for r in range(world)]
    assert np.count_nonzero(ref["status"]) == 1
    for k in ("mvn", "status", "free_energy"):
        got = np.concatenate([p[k] for p in parts], axis=-1)
        assert np.array_equal(got, ref[k], equal_nan=True), k


# ---- the same decomposition inside the engine: one process, several devices (fabber_vb_run_spatial_host_multi) ----
@pytest.mark.gpu
@pytest.mark.parametrize("typ,slabs", [("M", 2), ("P", 2), ("M", 3), ("M", 5)])
def test_engine_slabs_against_the_single_device_run(typ, slabs):
    """devices = the one GPU listed 2, 3, 5 times (a device copies to itself where two GPUs would copy to each
    other): z-slabs with ghost planes, pipelined first sweep, device-to-device hand-over of the boundary planes -
    every output identical to the one-device run, first- and second-neighbour priors"""
    h, sp, y = spatial_problem(typ)
    ref = hiplib.run_spatial_host(h, sp, y)
    got = hiplib.run_spatial_host(h, sp, y, devices=[0] * slabs)
    assert np.all(got["status"] == 0) and np.all(got["iterations"] == 8)
    for k in ("mvn", "status", "iterations"):
        assert np.array_equal(got[k], ref[k]), (typ, slabs, k)


@pytest.mark.gpu
@pytest.mark.parametrize("typ", ["M", "P"])
def test_resident_multi_device_run_can_be_repeated(typ):
    """fabber_vb_spatial_multi_*: the slabs' data stay on their devices, run() is a complete run (geometry, set-up,
    iterations, packing) and can be called again - what bench.py --workload c5 --gpus N times"""
    h, sp, y = spatial_problem(typ)
    ref = hiplib.run_spatial_host(h, sp, y)
    m = hiplib.SpatialMultiRun(h, sp, y, [0, 0, 0])
    for _ in range(2):
        m.run()
        got = m.results()
        n, route = m.slabs()
        assert n == 3 and route == "all slabs sweep together"
        for k in ("mvn", "status", "iterations"):
            assert np.array_equal(got[k], ref[k]), (typ, k)
    m.close()


@pytest.mark.gpu
def test_an_inbox_that_never_arrives_ends_in_the_level_chunk_pipeline():
    """The hand-over between slabs is a wait for a self-validating granule. If one never arrives - here: the slab
    below is left unlinked (fabber_vb_test_unlink_slab_pair), so the lowest plane of the slab above polls stale inboxes
    - ONE wait runs to its limit and raises the flag every other wait looks at, the sweep kernels drain, and the run is
    repeated as the level-chunk pipeline: the one-device bits, in seconds, not a hang"""
    import time
    h, sp, y = spatial_problem("M")
    ref = hiplib.run_spatial_host(h, sp, y)
    m = hiplib.SpatialMultiRun(h, sp, y, [0, 0, 0])
    hiplib.test_unlink_slab_pair(1)
    t0 = time.perf_counter()
    m.run()
    took = time.perf_counter() - t0
    got = m.results()
    n, route = m.slabs()
    assert n == 3 and route == "level-chunk pipeline"
    for k in ("mvn", "status", "iterations"):
        assert np.array_equal(got[k], ref[k]), k
    assert took < 60, took
    m.run()  # (the hook held for one attempt)
    assert m.slabs()[1] == "all slabs sweep together"
    assert np.array_equal(m.results()["mvn"], ref["mvn"])
    m.close()


@pytest.mark.gpu
def test_engine_slabs_with_free_energy_ard_and_a_failing_voxel():
    h, sp, y = ard_problem()
    ref = hiplib.run_spatial_host(h, sp, y)
    got = hiplib.run_spatial_host(h, sp, y, devices=[0, 0])
    assert np.count_nonzero(ref["status"]) == 1
    for k in ("mvn", "status", "free_energy", "iterations"):
        assert np.array_equal(got[k], ref[k], equal_nan=True), k
    # "all" on the one-GPU box is one slab: the single-device entry point itself
    alone = hiplib.run_spatial_host(h, sp, y, devices="all")
    assert np.array_equal(alone["mvn"], ref["mvn"], equal_nan=True)


@pytest.mark.gpu
def test_devices_option_with_spatialvb():
    """`devices=0,0` through fabber_dorun with method=spatialvb: the images of the run without it"""
    from fabber_core_amd import fabber
    rng = np.random.default_rng(8)
    shape, T = (7, 6, 8), 16
    t = np.arange(1, T + 1)
    c0 = 2.0 + np.sin(np.arange(shape[0])[:, None, None] / 2.0) + np.zeros(shape)
    data = (c0[..., None] + 0.3 * t + rng.normal(0, 0.2, shape + (T,))).astype(np.float32)
    opts = {"model": "poly", "degree": 1, "noise": "white", "method": "spatialvb", "max-iterations": 5, "param-spatial-priors": "MN",
            "save-mean": True, "save-mvn": True}
    one = fabber.run(data, opts)
    two = fabber.run(data, dict(opts, devices="0,0", **{"spatial-slabs": True}))
    assert "cut into z-slabs" in two["log"]
    for k in ("finalMVN", "mean_c0", "mean_c1"):
        assert np.array_equal(one[k], two[k]), k
    # devices= alone shards voxelwise VB: a spatial run stays on one device (the faster choice) and says so
    plain = fabber.run(data, dict(opts, devices="0,0"))
    assert "cut into z-slabs" not in plain["log"] and "uses one device" in plain["log"]
    assert np.array_equal(one["finalMVN"], plain["finalMVN"])


@pytest.mark.gpu
@pytest.mark.parametrize("typ", ["M", "P"])
def test_engine_slabs_on_a_very_unbalanced_mask(typ):
    """Plane voxel counts 1, 1, 1, 1, 1, 1, 60, 60 (round 2: the voxel-balanced cut ran off the end of the plane list
    or left a slab thinner than the halo of second-neighbour priors and the run was refused): every slab keeps its
    halo, and what cannot be cut runs on fewer slabs - the single-device result either way"""
    pts = [(0, 0, z) for z in range(6)] + [(x, y, z) for z in (6, 7) for y in range(6) for x in range(10)]
    coords = np.array(pts, dtype=np.int32).T.copy()
    V, T = coords.shape[1], 20
    rng = np.random.default_rng(31)
    t = np.arange(1, T + 1.0)
    y = 2.0 + 0.1 * coords[0][None, :] + 0.3 * t[:, None] + rng.normal(0, 0.2, (T, V))
    h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=1, max_iterations=5, param_overrides={"c0": dict(type=typ)})
    sp = vbabi.SpatialHolder(coords)
    ref = hiplib.run_spatial_host(h, sp, y)
    for slabs in (2, 3, 4):
        got = hiplib.run_spatial_host(h, sp, y, devices=[0] * slabs)
        assert np.array_equal(got["mvn"], ref["mvn"]) and np.array_equal(got["status"], ref["status"]), (typ, slabs)


@pytest.mark.gpu
@pytest.mark.parametrize("noise_kw", [dict(noise=vbabi.NOISE_AR1), dict(noise_pattern="12"),
                                      dict(noise=vbabi.NOISE_AR1, num_echoes=2, ar_cross_terms="dual")])
def test_engine_slabs_under_other_noise_models(noise_kw):
    """the slab driver runs the same kernels: AR(1) noise and two noise precisions, bit for bit the one-device run"""
    coords = masked_coords((8, 7, 10), seed=41, keep=0.9)
    V, T = coords.shape[1], 30
    rng = np.random.default_rng(42)
    t = np.arange(1, T + 1.0)
    y = 2.0 + np.sin(coords[0] / 2.0)[None, :] + 0.3 * t[:, None] + rng.normal(0, 0.3, (T, V))
    h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=1, max_iterations=5, need_f=True, param_overrides={"c0": dict(type="M")}, **noise_kw)
    sp = vbabi.SpatialHolder(coords)
    ref = hiplib.run_spatial_host(h, sp, y)
    got = hiplib.run_spatial_host(h, sp, y, devices=[0, 0, 0])
    for k in ("mvn", "status", "free_energy"):
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("slabs", [4, 8])
def test_engine_slabs_at_a_size_where_the_streams_overlap(slabs, monkeypatch):
    """64^3 bi-exponential volume (configs[4]'s model and prior), 3 iterations: with this much work in flight on
    every slab's stream a hand-over that is not ordered against them shows (it did: hipMemcpyPeer does not wait
    for non-blocking streams); also with the level range cut into chunks of 10"""
    import cases
    h, coords, y, _ = cases.c5_problem((64, 64, 64), max_iterations=3)
    sp = vbabi.SpatialHolder(coords)
    ref = hiplib.run_spatial_host(h, sp, y)
    got = hiplib.run_spatial_host(h, sp, y, devices=[0] * slabs)
    assert np.array_equal(got["mvn"], ref["mvn"], equal_nan=True) and np.array_equal(got["status"], ref["status"])
    monkeypatch.setenv("FVB_SPATIAL_CHUNK_LEVELS", "10")
    got = hiplib.run_spatial_host(h, sp, y, devices=[0] * slabs)
    assert np.array_equal(got["mvn"], ref["mvn"], equal_nan=True)
