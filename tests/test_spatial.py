"""Spatial VB (Vb::DoCalculationsSpatial, SpatialPrior, Vb::CalcNeighbours): oracle pinned on CPU,
HIP path against the oracle on the GPU. BASELINE config 5 model: bi-exponential + MRF prior (type
M) on a parameter; here at oracle-sized volumes with irregular masks."""
import numpy as np
import pytest

import golden_utils as gu
import oracle
import parity
from fabber_core_amd import hiplib, vbabi

gpu = pytest.mark.gpu
needs_lib = pytest.mark.skipif(not hiplib.available(), reason="libfabber_vb_hip.so not built")


def masked_volume(shape, seed, keep=0.85):
    rng = np.random.default_rng(seed)
    mask = rng.random(shape) < keep
    return mask, vbabi.grid_coords(shape, mask)


def smooth_exp_data(coords, T, dt, seed, noise_sd=0.1):
    rng = np.random.default_rng(seed)
    t = np.arange(T) * dt
    amp = 1.0 + 0.3 * np.sin(coords[0] / 3.0) * np.cos(coords[1] / 4.0) + 0.1 * np.sin(coords[2] / 2.0)
    return amp, amp[None, :] * np.exp(-1.0 * t[:, None]) + rng.normal(0, noise_sd, (T, coords.shape[1]))


# ---------------------------------------------------------------------------------------------
# CPU
# ---------------------------------------------------------------------------------------------
def test_neighbour_lists_on_a_full_grid():
    """3x3x3 block: corner 3, edge 4, face 5, centre 6 first neighbours; second neighbours with
    duplicates excluding the voxel itself (inference_vb.cc:830-964)."""
    coords = vbabi.grid_coords((3, 3, 3))
    nn, nn2, n2c = oracle.calc_neighbours(coords)
    counts = (nn > 0).sum(axis=1)
    assert sorted(np.bincount(counts)[3:].tolist()) == sorted([8, 12, 6, 1])
    centre = 13
    assert counts[centre] == 6 and set(nn[centre]) == {15, 13, 17, 11, 23, 5}
    assert list(nn[0][:3]) == [2, 4, 10]          # +x, +y, +z in the reference's order (1-based)
    assert n2c[centre] == 6 * 4                   # each face voxel has 4 other neighbours
    # 2D: no z neighbours
    nn2d, _, _ = oracle.calc_neighbours(coords, spatial_dims=2)
    assert (nn2d[centre] > 0).sum() == 4


def _cube(n, origin=0):
    c = vbabi.grid_coords((n, n, n))
    return (c + origin).astype(np.int32)


def _first_neighbour_sets(coords, dims=3):
    """Neighbour sets (1-based ids) from the oracle and from the driver's table; they must agree."""
    nn, nn2, n2c = oracle.calc_neighbours(coords, dims)
    ref = [set(row[row > 0].tolist()) for row in nn]
    if hiplib.available():
        drv = hiplib.neighbours(coords, dims)
        assert [set((row[row >= 0] + 1).tolist()) for row in drv] == ref
    return ref, nn2, n2c


def test_reference_calc_neighbours_cases():
    """test/test_spatialvb.cc:81-582, every CalcNeighbours case of the reference's own suite."""
    # one voxel, at 1 and at 0 (:81-105)
    for at in (1, 0):
        ref, _, n2c = _first_neighbour_sets(np.full((3, 1), at, dtype=np.int32))
        assert ref == [set()] and n2c[0] == 0
    # five voxels along x, y, z (:107-187)
    for axis in range(3):
        c = np.ones((3, 5), dtype=np.int32)
        c[axis] = np.arange(1, 6)
        ref, _, _ = _first_neighbour_sets(c)
        assert [len(s) for s in ref] == [1, 2, 2, 2, 1]
    # 5x5x5 cube with co-ordinates from 0 (:189-247) and from 1 (:460-518)
    n = 5
    for origin in (0, 1):
        ref, nn2, n2c = _first_neighbour_sets(_cube(n, origin))
        v = 1
        for z in range(n):
            for y in range(n):
                for x in range(n):
                    want = set()
                    if x != 0: want.add(v - 1)
                    if x != n - 1: want.add(v + 1)
                    if y != 0: want.add(v - n)
                    if y != n - 1: want.add(v + n)
                    if z != 0: want.add(v - n * n)
                    if z != n - 1: want.add(v + n * n)
                    assert ref[v - 1] == want
                    if origin == 0:  # second neighbours, duplicates kept (:365-457)
                        exp = []
                        if x >= 2: exp.append(v - 2)
                        if x <= n - 3: exp.append(v + 2)
                        if y >= 2: exp.append(v - 2 * n)
                        if y <= n - 3: exp.append(v + 2 * n)
                        if z >= 2: exp.append(v - 2 * n * n)
                        if z <= n - 3: exp.append(v + 2 * n * n)
                        for _ in range(2):
                            for (a, da), (b, db) in (((x, 1), (y, n)), ((x, 1), (z, n * n)), ((y, n), (z, n * n))):
                                if a >= 1 and b >= 1: exp.append(v - da - db)
                                if a >= 1 and b <= n - 2: exp.append(v - da + db)
                                if a <= n - 2 and b >= 1: exp.append(v + da - db)
                                if a <= n - 2 and b <= n - 2: exp.append(v + da + db)
                        assert n2c[v - 1] == len(exp)
                        assert sorted(nn2[v - 1][:n2c[v - 1]].tolist()) == sorted(exp)
                    v += 1
    # reduced spatial dimensions on the cube (:251-363)
    for dims, per_axis in ((1, 1), (2, 2)):
        ref, _, _ = _first_neighbour_sets(_cube(n), dims)
        v = 1
        for z in range(n):
            for y in range(n):
                for x in range(n):
                    want = set()
                    if x != 0: want.add(v - 1)
                    if x != n - 1: want.add(v + 1)
                    if dims >= 2 and y != 0: want.add(v - n)
                    if dims >= 2 and y != n - 1: want.add(v + n)
                    assert ref[v - 1] == want
                    v += 1
    # irregular five-voxel volume (:521-582)
    c = np.array([[1, 2, 1, 2, 1], [1, 1, 2, 2, 1], [1, 1, 1, 1, 2]], dtype=np.int32)
    ref, _, n2c = _first_neighbour_sets(c)
    assert [len(s) for s in ref] == [3, 2, 2, 2, 1]
    assert n2c.tolist() == [2, 3, 3, 2, 2]


def test_no_wrap_around_between_rows():
    """The last voxel of a row and the first of the next have consecutive offsets but are not
    neighbours (wrap-around test, inference_vb.cc:906-925)."""
    coords = vbabi.grid_coords((4, 3, 1))
    nn, _, _ = oracle.calc_neighbours(coords)
    assert 5 not in nn[3]  # voxel 4 (x=3,y=0) vs voxel 5 (x=0,y=1)
    assert set(nn[3][nn[3] > 0]) == {3, 8}


def test_misordered_coordinates_are_rejected():
    coords = vbabi.grid_coords((3, 3, 1))[:, ::-1].copy()
    with pytest.raises(RuntimeError, match="correct order"):
        oracle.calc_neighbours(coords)


@needs_lib
@pytest.mark.parametrize("dims", [3, 2, 1])
def test_driver_neighbour_table_matches_reference(dims):
    mask, coords = masked_volume((9, 7, 5), seed=dims)
    ref, _, _ = oracle.calc_neighbours(coords, dims)
    got = hiplib.neighbours(coords, dims)
    assert np.array_equal(got + 1, ref)  # same neighbours in the same order (reference is 1-based, 0 = none)


@needs_lib
def test_driver_neighbour_table_sparse_and_large_geometries():
    """The driver answers 'which voxel sits at offset pos + delta' from a dense map when the mask
    fills its bounding box reasonably and by the reference's binary search otherwise."""
    rng = np.random.default_rng(0)
    # a few small blobs scattered through a 400^3 box: bounding span >> 64 V -> binary search
    pts = set()
    for cx, cy, cz in rng.integers(5, 395, (6, 3)):
        for dx, dy, dz in np.ndindex(3, 3, 3):
            if rng.random() < 0.8:
                pts.add((int(cx + dx), int(cy + dy), int(cz + dz)))
    pts = sorted(pts, key=lambda p: (p[2], p[1], p[0]))
    coords = np.array(pts, dtype=np.int32).T.copy()
    ref, _, _ = oracle.calc_neighbours(coords, 3)
    assert np.array_equal(hiplib.neighbours(coords, 3) + 1, ref)
    assert (ref > 0).sum() > 100
    # a full 40^3 block through the dense map
    coords = vbabi.grid_coords((40, 40, 40))
    ref, _, _ = oracle.calc_neighbours(coords, 3)
    assert np.array_equal(hiplib.neighbours(coords, 3) + 1, ref)


def test_oracle_spatial_loop_reproduces_reference_spatialvb_output():
    """test/outdata_linear_spatialvb (method=spatialvb, all-N priors): the spatial loop's fixed
    point, replayed from data with the same sufficient statistics."""
    ref = gu.load_reference_outdata()
    J, mvn = ref["linear_design"], ref["linear_spatialvb/finalMVN"].astype(np.float64)
    cov, means = gu.unpack(mvn, 5)
    y = gu.data_with_same_sufficient_statistics(J, cov, means)
    idx, (nx, ny, nz) = ref["mask_index"], [int(s) for s in ref["mask_shape"]]
    coords = np.stack([idx % nx, (idx // nx) % ny, idx // (nx * ny)]).astype(np.int32)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, 147, 106, design=J)
    r = oracle.run_spatial(h, vbabi.SpatialHolder(coords), y)
    gc, gm = gu.unpack(r["mvn"], 5)
    sd = np.sqrt(np.einsum("vii->vi", cov))
    assert np.max(np.abs(gm - means) / np.maximum(sd, np.abs(means))) < 1e-5
    assert np.max(np.abs(gc - cov) / (sd[:, :, None] * sd[:, None, :])) < 1e-5


@pytest.mark.parametrize("dims", [3, 2, 1])
@pytest.mark.parametrize("typ", ["P", "p"])
def test_second_neighbour_prior_types_as_the_reference_codes_them(typ, dims):
    """priors.cc:455 `double rec = 1 / (8*nn - nn2);` divides two ints: 0 for every reachable neighbourhood (nn2 <= 5 nn).
    As coded spatial_mean is 0, so the prior mean of types P and p never reads a neighbour's value:
      P: mu0 = prec0 mean0 / (prec0 + aK (nn^2 + nn)),  p: mu0 = prec0 mean0 / (aK (4 d^2 + 2 d))  (42 aK for d = 3)
    while types M and m do smooth (mean of the neighbours). The precisions are what the formula says either way."""
    mask, coords = masked_volume((6, 5, 4), seed=2)
    V = coords.shape[1]
    rng = np.random.default_rng(3)
    means = rng.normal(2.0, 1.0, V)
    mean0, prec0, aK = 0.7, 1e-3, 0.37
    nn_ids, nn2_ids, n2c = oracle.calc_neighbours(coords, dims)
    nn = (nn_ids > 0).sum(axis=1)
    assert np.all(8 * nn[nn > 0] - n2c[nn > 0] >= 3)  # the divisor of :455: its integer reciprocal is 0
    pm, pp = oracle.spatial_prior_apply(coords, typ, means, mean0, prec0, aK, spatial_dims=dims)
    if typ == "P":
        want_prec = prec0 + aK * (nn * nn + nn)
        want_mean = prec0 * mean0 / want_prec
    else:
        want_prec = np.full(V, aK * (4 * dims * dims + 2 * dims))
        want_mean = prec0 * mean0 / want_prec
    assert np.allclose(pp, want_prec, rtol=1e-14, atol=0)
    assert np.allclose(pm, want_mean, rtol=1e-13, atol=0)
    # not a function of the neighbours' values ...
    pm2, _ = oracle.spatial_prior_apply(coords, typ, rng.normal(-5.0, 3.0, V), mean0, prec0, aK, spatial_dims=dims)
    assert np.array_equal(pm, pm2)
    # ... only of whether they are finite: 0 x inf = NaN reaches the first and second neighbours of a bad voxel
    bad = V // 2
    poisoned = means.copy()
    poisoned[bad] = np.inf
    pm3, _ = oracle.spatial_prior_apply(coords, typ, poisoned, mean0, prec0, aK, spatial_dims=dims)
    reach = set((nn_ids[bad][nn_ids[bad] > 0] - 1).tolist()) | set((nn2_ids[bad][:n2c[bad]] - 1).tolist())
    assert set(np.flatnonzero(np.isnan(pm3)).tolist()) == reach
    # types M / m next to them: the mean of the neighbours
    pmM, ppM = oracle.spatial_prior_apply(coords, "M", means, mean0, prec0, aK, spatial_dims=dims)
    for v in range(V):
        ids = nn_ids[v][nn_ids[v] > 0] - 1
        if len(ids):
            sp = aK * (len(ids) + 1e-8)
            assert np.isclose(pmM[v], sp / (prec0 + sp) * means[ids].mean(), rtol=1e-12)


@pytest.mark.parametrize("typ", ["P", "p"])
def test_oracle_second_neighbour_types_do_not_smooth(typ):
    """The run-level consequence of the integer division: a P / p prior pulls towards prec0 mean0 / precision, i.e. it
    only shrinks. With types M / m the fitted map is closer to the smooth truth than the voxelwise fit; with P / p it is
    the voxelwise fit shrunk (never closer to the truth than M)."""
    mask, coords = masked_volume((8, 7, 6), seed=0)
    V = coords.shape[1]
    amp, y = smooth_exp_data(coords, 50, 0.04, seed=1)
    run = lambda t: oracle.run_spatial(vbabi.build_config(vbabi.MODEL_EXP, V, 50, num_exps=1, dt=0.04, max_iterations=10,
                                                         param_overrides={"amp1": dict(type=t)}), vbabi.SpatialHolder(coords), y)
    r, rM = run(typ), run("M")
    rmse = lambda res: np.sqrt(np.mean((np.exp(res["mvn"][6]) - amp) ** 2))
    assert np.isfinite(r["mvn"]).all()
    assert rmse(rM) < rmse(r)


@pytest.mark.parametrize("typ", ["M", "m"])
def test_oracle_spatial_prior_smooths(typ):
    mask, coords = masked_volume((8, 7, 6), seed=0)
    V = coords.shape[1]
    amp, y = smooth_exp_data(coords, 50, 0.04, seed=1)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 50, num_exps=1, dt=0.04, max_iterations=10, param_overrides={"amp1": dict(type=typ)})
    r = oracle.run_spatial(h, vbabi.SpatialHolder(coords), y)
    hv = vbabi.build_config(vbabi.MODEL_EXP, V, 50, num_exps=1, dt=0.04, max_iterations=10)
    rv = oracle.run(hv, y)
    rmse = lambda res: np.sqrt(np.mean((np.exp(res["mvn"][6]) - amp) ** 2))
    assert np.isfinite(r["mvn"]).all()
    assert rmse(r) < 0.95 * rmse(rv)


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------
def spatial_check(h, sp, y, what, **kw):
    cpu = oracle.run_spatial(h, sp, y)
    got = hiplib.run_spatial_host(h, sp, y)
    cpu2 = oracle.run_spatial_fma(h, sp, y)
    for r in (cpu, cpu2):
        r.setdefault("f_history_len", np.zeros(h.cfg.n_voxels, dtype=np.int32))
    return parity.strict(h, cpu, got, what=what, cpu2=cpu2, **kw)


@gpu
@pytest.mark.parametrize("typ", ["M", "m", "P", "p"])
def test_spatial_prior_types_single_exponential(typ):
    mask, coords = masked_volume((11, 9, 7), seed=3)
    V = coords.shape[1]
    _, y = smooth_exp_data(coords, 50, 0.04, seed=4)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 50, num_exps=1, dt=0.04, max_iterations=8, param_overrides={"amp1": dict(type=typ)})
    spatial_check(h, vbabi.SpatialHolder(coords), y, "spatial " + typ)


@gpu
def test_spatial_options_dims_speed_first_iteration_and_free_energy():
    mask, coords = masked_volume((10, 8, 6), seed=5)
    V = coords.shape[1]
    rng = np.random.default_rng(6)
    t = np.arange(1, 21.0)
    c0 = 2.0 + np.sin(coords[0] / 2.0)
    y = c0[None, :] + 0.3 * t[:, None] + rng.normal(0, 0.2, (20, V))
    h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=1, max_iterations=6, need_f=True,
                           param_overrides={"c0": dict(type="M"), "c1": dict(type="A")})
    spatial_check(h, vbabi.SpatialHolder(coords, spatial_dims=2, update_first_iter=True), y, "dims2+first-iter+ARD", check_f=True)
    h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=1, max_iterations=6,
                           param_overrides={"c0": dict(type="P"), "c1": dict(type="m")})
    spatial_check(h, vbabi.SpatialHolder(coords, spatial_speed=1.5, q1=5.0, q2=2.0), y, "speed+q1q2, two spatial params")


def failing_voxel_problem(typ, need_f=True):
    mask, coords = masked_volume((9, 8, 6), seed=11, keep=0.9)
    V = coords.shape[1]
    _, y = smooth_exp_data(coords, 50, 0.04, seed=12)
    bad = [V // 2, V // 2 + 1, V - 1]  # two adjacent interior voxels and the LAST voxel
    for v in bad:
        y[7, v] = np.nan  # non-finite means after the first UpdateTheta -> ReCentre throws -> IgnoreVoxel
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 50, num_exps=1, dt=0.04, max_iterations=6, need_f=need_f,
                           param_overrides={"amp1": dict(type=typ)})
    return h, vbabi.SpatialHolder(coords), y, bad


@pytest.mark.parametrize("typ", ["M", "P"])
def test_oracle_ignores_failed_voxels_and_their_neighbours_carry_on(typ):
    """Vb::IgnoreVoxel (inference_vb.cc:266-297): a voxel that fails is dropped from its
    neighbours' lists and from the a_K sums; everything else finishes."""
    h, sp, y, bad = failing_voxel_problem(typ)
    r = oracle.run_spatial(h, sp, y)
    assert sorted(np.flatnonzero(r["status"]).tolist()) == sorted(bad)
    ok = r["status"] == 0
    assert np.isfinite(r["mvn"][:, ok]).all() and np.isfinite(r["free_energy"][ok]).all()


@gpu
@pytest.mark.parametrize("typ", ["M", "m", "P", "p"])
def test_failed_voxels_are_ignored_as_in_the_reference(typ):
    """With F evaluated, the non-finite sample is caught by the first CalculateF of the sweep,
    before the voxel's means change: exactly the three voxels fail."""
    h, sp, y, bad = failing_voxel_problem(typ)
    # (type p: 1.5e-6 on one mean where the two CPU builds differ by 2.0e-6 - the kernels contract multiply-adds as the
    # FMA build of the oracle does)
    spatial_check(h, sp, y, "IgnoreVoxel " + typ, check_f=True, allow_floor=(typ == "p"))
    got = hiplib.run_spatial_host(h, sp, y)
    assert sorted(np.flatnonzero(got["status"]).tolist()) == sorted(bad)


@gpu
def test_without_f_a_failure_spreads_along_the_sweep_as_in_the_reference():
    """Without F nothing looks at the voxel between UpdateTheta (non-finite means) and ReCentre in
    the second sweep: every later neighbour reads those means in the meantime. The level-ordered
    sweep reproduces the reference's index-ordered cascade voxel for voxel."""
    h, sp, y, bad = failing_voxel_problem("M", need_f=False)
    cpu = oracle.run_spatial(h, sp, y)
    assert set(bad) < set(np.flatnonzero(cpu["status"]).tolist())
    spatial_check(h, sp, y, "IgnoreVoxel cascade")


@gpu
def test_spatialvb_method_with_nonspatial_priors_reproduces_reference_output():
    ref = gu.load_reference_outdata()
    J, mvn = ref["linear_design"], ref["linear_spatialvb/finalMVN"].astype(np.float64)
    cov, means = gu.unpack(mvn, 5)
    y = gu.data_with_same_sufficient_statistics(J, cov, means)
    idx, (nx, ny, nz) = ref["mask_index"], [int(s) for s in ref["mask_shape"]]
    coords = np.stack([idx % nx, (idx // nx) % ny, idx // (nx * ny)]).astype(np.int32)
    h = vbabi.build_config(vbabi.MODEL_LINEAR, 147, 106, design=J)
    r = hiplib.run_spatial_host(h, vbabi.SpatialHolder(coords), y)
    gc, gm = gu.unpack(r["mvn"], 5)
    sd = np.sqrt(np.einsum("vii->vi", cov))
    assert np.all(r["status"] == 0)
    assert np.max(np.abs(gm - means) / np.maximum(sd, np.abs(means))) < 1e-3
    assert np.max(np.abs(gc - cov) / (sd[:, :, None] * sd[:, None, :])) < 1e-3


@gpu
@pytest.mark.parametrize("route", ["default", "per-level", "two slabs"])
def test_c5_error_against_the_ground_truth(route, monkeypatch):
    """BASELINE configs[4] - bi-exponential + MRF prior (type M) on amp1, 10 iterations, F evaluated - on the 16 x 16 x 12
    block whose exact result is committed (tests/golden/c5_truth_binary128.npz: the oracle's spatial loop in IEEE
    binary128). After 10 iterations the voxels are still inside the chaotic phase of the fit (two CPU builds are
    within 1e-4 of the truth on 1 % of them), so per-voxel parity with a CPU build means nothing here; the kernels'
    error against the truth - posterior and free energy - must be no worse than the worse CPU build's, no slack.
    Routes: the split first sweep (slab form), the per-level launches, the engine's slab driver."""
    truth, (h, sp, y) = parity.load_c5_truth()
    if route == "per-level":
        monkeypatch.setenv("FVB_SPATIAL_PER_LEVEL", "1")
    devices = [0, 0] if route == "two slabs" else None
    got = hiplib.run_spatial_host(h, sp, y, devices=devices)
    gpu_s = parity.truth_stats(h, truth, got, with_f=True)
    c1, c2 = (parity.truth_stats(h, truth, r(h, sp, y), with_f=True) for r in (oracle.run_spatial, oracle.run_spatial_fma))
    print("C5 vs binary128 truth [%s]: gpu %s | cpu %s | cpu_fma %s" % (route, gpu_s, c1, c2))
    # median, 75th percentile, failed share, F where the posterior is the truth's: no slack. The 90th / 99th
    # percentile after 10 iterations are voxels 0.1 - 2 posterior sd off in every build (the run stops inside the
    # chaotic phase): 10 % on those two.
    parity.no_worse_than_the_cpu_builds(gpu_s, c1, c2, what="C5 " + route, with_f=True, median_factor=1.0, tail_factor=1.1)
    # ... and on the way there: the median error of the means after 1 ... 9 iterations, as the C3 test (x 1.1)
    import make_c5_truth as mt
    for k, it in enumerate(truth["its"]):
        hk, _, _ = mt.problem(max_iterations=int(it))
        g = parity.truth_trace_stats(hk, truth["trace_means"][k], hiplib.run_spatial_host(hk, sp, y, devices=devices))["median"]
        c = max(parity.truth_trace_stats(hk, truth["trace_means"][k], r(hk, sp, y))["median"] for r in (oracle.run_spatial, oracle.run_spatial_fma))
        assert g <= 1.1 * c, (route, it, g, c)


@gpu
def test_c5_biexponential_mrf_population():
    """BASELINE config 5 model on a 16x14x12 masked block: bi-exponential with an MRF prior (M) on
    amp1. The bi-exponential fit is chaotic per voxel, so population parity."""
    mask, coords = masked_volume((16, 14, 12), seed=7, keep=0.9)
    V = coords.shape[1]
    rng = np.random.default_rng(8)
    t = np.arange(100) * 0.02
    amp1 = 0.75 + 0.25 * np.sin(coords[0] / 4.0) * np.cos(coords[2] / 3.0)
    y = amp1[None, :] * np.exp(-1.0 * t[:, None]) + 0.5 * np.exp(-6.0 * t[:, None]) + rng.normal(0, 0.1, (100, V))
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 100, num_exps=2, dt=0.02, max_iterations=10, param_overrides={"amp1": dict(type="M")})
    sp = vbabi.SpatialHolder(coords)
    cpu, cpu2, got = oracle.run_spatial(h, sp, y), oracle.run_spatial_fma(h, sp, y), hiplib.run_spatial_host(h, sp, y)
    floor = parity.population_stats(h, cpu, cpu2)
    parity.population(h, cpu, got, floor, what="C5")
    calls = []
    hiplib.run_spatial_host(h, sp, y, progress_cb=lambda i, n: calls.append((i, n)))
    assert calls == [(i, 10) for i in range(10)]


@gpu
@pytest.mark.parametrize("dims", [3, 2])
def test_neighbour_table_built_on_the_device_gives_the_same_run(dims, monkeypatch):
    """The driver builds the first-neighbour table in three kernels when the geometry allows
    (non-negative co-ordinates, reasonably dense mask) and on the host otherwise; the host table
    is checked against the reference's lists above. Same run bit for bit either way."""
    mask, coords = masked_volume((12, 10, 8), seed=21, keep=0.8)
    V = coords.shape[1]
    _, y = smooth_exp_data(coords, 40, 0.04, seed=22)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 40, num_exps=1, dt=0.04, max_iterations=5, param_overrides={"amp1": dict(type="P")})
    sp = vbabi.SpatialHolder(coords, spatial_dims=dims)
    dev = hiplib.run_spatial_host(h, sp, y)
    monkeypatch.setenv("FVB_SPATIAL_HOST_GEOMETRY", "1")
    host = hiplib.run_spatial_host(h, sp, y)
    assert np.array_equal(dev["mvn"], host["mvn"]) and np.array_equal(dev["status"], host["status"])
    # a geometry the device path declines (two slabs 10 000 planes apart: the offset -> voxel map
    # would be far larger than the mask) takes the host path by itself, with the same lists
    monkeypatch.delenv("FVB_SPATIAL_HOST_GEOMETRY")
    far = coords.copy()
    far[2, coords[2] >= 4] += 10000
    a = hiplib.run_spatial_host(h, vbabi.SpatialHolder(far, spatial_dims=dims), y)
    monkeypatch.setenv("FVB_SPATIAL_HOST_GEOMETRY", "1")
    b = hiplib.run_spatial_host(h, vbabi.SpatialHolder(far, spatial_dims=dims), y)
    assert np.array_equal(a["mvn"], b["mvn"])
    assert np.array_equal(a["mvn"], host["mvn"]) == (dims == 2)     # (in 3 dimensions the cut removed neighbours)


@gpu
def test_misordered_coordinates_are_rejected_by_the_device_scan():
    coords = vbabi.grid_coords((4, 4, 2))[:, ::-1].copy()
    h = vbabi.build_config(vbabi.MODEL_POLY, 32, 10, degree=0, param_overrides={"c0": dict(type="M")})
    with pytest.raises(hiplib.HipEngineError, match="correct order"):
        hiplib.run_spatial_host(h, vbabi.SpatialHolder(coords), np.ones((10, 32), dtype=np.float32))


@gpu
def test_level_order_built_by_several_host_threads_is_the_same(monkeypatch):
    """Volumes from 262 144 voxels up sort their sweep levels with a few host threads (stable
    counting sort with per-thread histograms): forced here on a small volume."""
    mask, coords = masked_volume((13, 11, 9), seed=31, keep=0.85)
    V = coords.shape[1]
    _, y = smooth_exp_data(coords, 30, 0.04, seed=32)
    h = vbabi.build_config(vbabi.MODEL_EXP, V, 30, num_exps=1, dt=0.04, max_iterations=4, param_overrides={"amp1": dict(type="P")})
    sp = vbabi.SpatialHolder(coords)
    one = hiplib.run_spatial_host(h, sp, y)
    for n in ("3", "7"):
        monkeypatch.setenv("FVB_SPATIAL_HOST_THREADS", n)
        many = hiplib.run_spatial_host(h, sp, y)
        assert np.array_equal(one["mvn"], many["mvn"]) and np.array_equal(one["status"], many["status"])


@gpu
@pytest.mark.parametrize("case", ["M exp", "m exp masked", "two spatial + ARD + F", "M + image prior, dims 2, speed", "failing voxel with F",
                                  "NaN cascade without F"])
def test_split_first_sweep_is_the_per_level_sweep_bit_for_bit(case, monkeypatch):
    """Whole-volume runs take the split first sweep (vb_spatial.h: one parallel launch for everything that does not
    wait for a neighbour, ONE persistent launch that walks the levels for the means of the parameters with a
    first-neighbour prior, the rest inside the second sweep's kernel); the per-level launches stay for
    host-evaluated models, for the level-chunk pipeline of several slabs, and as the fallback when a voxel fails
    during a sweep. Both evaluate eq (19)-(20) with the same operation sequence: every output must be identical."""
    rng = np.random.default_rng(41)
    sp_kw = {}
    if case == "M exp":
        mask, coords = masked_volume((13, 11, 9), seed=31, keep=1.0)
        _, y = smooth_exp_data(coords, 40, 0.04, seed=32)
        h = vbabi.build_config(vbabi.MODEL_EXP, coords.shape[1], 40, num_exps=1, dt=0.04, max_iterations=7, param_overrides={"amp1": dict(type="M")})
    elif case == "m exp masked":
        mask, coords = masked_volume((12, 10, 8), seed=33, keep=0.8)
        _, y = smooth_exp_data(coords, 40, 0.04, seed=34)
        h = vbabi.build_config(vbabi.MODEL_EXP, coords.shape[1], 40, num_exps=1, dt=0.04, max_iterations=6, param_overrides={"r1": dict(type="m")})
    elif case in ("two spatial + ARD + F", "M + image prior, dims 2, speed"):
        mask, coords = masked_volume((10, 8, 6), seed=5)
        V = coords.shape[1]
        t = np.arange(1, 21.0)
        c0 = 2.0 + np.sin(coords[0] / 2.0)
        y = c0[None, :] + 0.3 * t[:, None] + 0.01 * t[:, None] ** 2 + rng.normal(0, 0.2, (20, V))
        if case.startswith("two"):
            h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=2, max_iterations=6, need_f=True,
                                   param_overrides={"c0": dict(type="M"), "c1": dict(type="A"), "c2": dict(type="m")})
            sp_kw = dict(update_first_iter=True)
        else:
            img = rng.normal(0.3, 0.05, V)
            h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=2, max_iterations=6,
                                   param_overrides={"c0": dict(type="M"), "c1": dict(type="I", prec=4.0)}, image_priors={"c1": img})
            sp_kw = dict(spatial_dims=2, spatial_speed=1.5, q1=5.0, q2=2.0)
    elif case == "failing voxel with F":
        h, sp0, y, bad = failing_voxel_problem("M")
        coords = sp0.coords
    else:
        h, sp0, y, bad = failing_voxel_problem("M", need_f=False)
        coords = sp0.coords
    sp = vbabi.SpatialHolder(coords, **sp_kw)
    # the ordered part: workgroups that own z-slabs of 1 / 2 / 3 planes (previous level in LDS, inbox hand-over
    # between slabs; the default picks the thickness from the volume)
    forms = {}
    for name, env in (("default", {}), ("slabs of 1", {"FVB_SPATIAL_SLAB_DZ": "1"}), ("slabs of 2", {"FVB_SPATIAL_SLAB_DZ": "2"}),
                      ("slabs of 3", {"FVB_SPATIAL_SLAB_DZ": "3"}),
                      ("slab numbering on the host", {"FVB_SPATIAL_HOST_NUMBERING": "1"}),
                      ("prep kernel in index order", {"FVB_SPATIAL_PREP_LINEAR": "1"}),
                      ("geometry on the host", {"FVB_SPATIAL_HOST_GEOMETRY": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        forms[name] = hiplib.run_spatial_host(h, sp, y)
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("FVB_SPATIAL_PER_LEVEL", "1")
    per_level = hiplib.run_spatial_host(h, sp, y)
    for name, split in forms.items():
        for k in ("mvn", "status", "iterations", "free_energy"):
            assert np.array_equal(split[k], per_level[k], equal_nan=True), (case, name, k)


@gpu
@pytest.mark.parametrize("case", ["P exp", "p exp masked", "P + M + ARD + F", "p, dims 2, speed", "failing voxel with F", "NaN cascade without F",
                                  "P and p, nothing to sweep", "minus zero"])
def test_second_neighbour_types_in_the_split_form_are_the_per_level_sweep_bit_for_bit(case, monkeypatch):
    """Types P and p as the reference codes them (priors.cc:455: the MRF mean is 0 x a sum) read no neighbour's VALUE:
    the split form takes their prior mean as pcov x prec0 mean0 in the prep kernel and sweeps only the parameters of
    types M / m (none: no ordered launch at all). The per-level kernel evaluates the reference's expression as it
    stands - second-neighbour sums, levels x + 2y + 3z, NaN where a listed mean is not finite. Identical output,
    including the runs the split form hands back (a voxel failing during a sweep; a non-finite mean next to such a
    prior, which the reference spreads to first and second neighbours in voxel order)."""
    rng = np.random.default_rng(43)
    sp_kw = {}
    if case == "P exp":
        mask, coords = masked_volume((13, 11, 9), seed=31, keep=1.0)
        _, y = smooth_exp_data(coords, 40, 0.04, seed=32)
        h = vbabi.build_config(vbabi.MODEL_EXP, coords.shape[1], 40, num_exps=1, dt=0.04, max_iterations=7, param_overrides={"amp1": dict(type="P")})
    elif case == "p exp masked":
        mask, coords = masked_volume((12, 10, 8), seed=33, keep=0.8)
        _, y = smooth_exp_data(coords, 40, 0.04, seed=34)
        h = vbabi.build_config(vbabi.MODEL_EXP, coords.shape[1], 40, num_exps=1, dt=0.04, max_iterations=6, param_overrides={"r1": dict(type="p")})
    elif case in ("P + M + ARD + F", "p, dims 2, speed", "P and p, nothing to sweep", "minus zero"):
        mask, coords = masked_volume((10, 8, 6), seed=5)
        V = coords.shape[1]
        t = np.arange(1, 21.0)
        c0 = 2.0 + np.sin(coords[0] / 2.0)
        y = c0[None, :] + 0.3 * t[:, None] + 0.01 * t[:, None] ** 2 + rng.normal(0, 0.2, (20, V))
        if case.startswith("P +"):
            h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=2, max_iterations=6, need_f=True,
                                   param_overrides={"c0": dict(type="P"), "c1": dict(type="A"), "c2": dict(type="M")})
            sp_kw = dict(update_first_iter=True)
        elif case.startswith("P and p"):
            h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=2, max_iterations=6, need_f=True,
                                   param_overrides={"c0": dict(type="P", mean=1.5, prec=0.01), "c2": dict(type="p", mean=0.02, prec=3.0)})
        elif case == "minus zero":
            # prec0 x mean0 = -0: the sign of the reference's zero decides the sign of a zero prior mean - per-level launches
            h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=2, max_iterations=4, param_overrides={"c0": dict(type="P", mean=-0.0)})
        else:
            h = vbabi.build_config(vbabi.MODEL_POLY, V, 20, degree=2, max_iterations=6, param_overrides={"c0": dict(type="p")})
            sp_kw = dict(spatial_dims=2, spatial_speed=1.5, q1=5.0, q2=2.0)
    elif case == "failing voxel with F":
        h, sp0, y, bad = failing_voxel_problem("P")
        coords = sp0.coords
    else:
        h, sp0, y, bad = failing_voxel_problem("P", need_f=False)
        coords = sp0.coords
    sp = vbabi.SpatialHolder(coords, **sp_kw)
    forms = {}
    for name, env in (("default", {}), ("geometry on the host", {"FVB_SPATIAL_HOST_GEOMETRY": "1"}), ("slabs of 2", {"FVB_SPATIAL_SLAB_DZ": "2"}),
                      ("slab numbering on the host", {"FVB_SPATIAL_HOST_NUMBERING": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        forms[name] = hiplib.run_spatial_host(h, sp, y)
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("FVB_SPATIAL_PER_LEVEL", "1")
    per_level = hiplib.run_spatial_host(h, sp, y)
    for name, split in forms.items():
        for k in ("mvn", "status", "iterations", "free_energy"):
            assert np.array_equal(split[k], per_level[k], equal_nan=True), (case, name, k)
    if case == "NaN cascade without F":
        # (the reference's cascade: more voxels than the three with a bad sample end up ignored)
        assert set(bad) < set(np.flatnonzero(per_level["status"]).tolist())
    monkeypatch.delenv("FVB_SPATIAL_PER_LEVEL")
    cpu = oracle.run_spatial(h, sp, y)
    cpu.setdefault("f_history_len", np.zeros(h.cfg.n_voxels, dtype=np.int32))
    parity.strict(h, cpu, forms["default"], what="P/p as coded: " + case, cpu2=oracle.run_spatial_fma(h, sp, y), check_f=bool(h.cfg.need_f), allow_floor=True)


@gpu
def test_slab_sweep_with_runs_longer_than_a_workgroup(monkeypatch):
    """a 40 x 40 x 3 volume swept as ONE slab of three planes: its longest run (the voxels of one level) has more
    than 64 voxels, and with FVB_SPATIAL_SLAB_WIDTH=64 the group's lanes take several voxels each"""
    mask, coords = masked_volume((40, 40, 3), seed=35, keep=0.97)
    _, y = smooth_exp_data(coords, 30, 0.04, seed=36)
    h = vbabi.build_config(vbabi.MODEL_EXP, coords.shape[1], 30, num_exps=1, dt=0.04, max_iterations=4, param_overrides={"amp1": dict(type="M")})
    sp = vbabi.SpatialHolder(coords)
    monkeypatch.setenv("FVB_SPATIAL_SLAB_DZ", "3")
    monkeypatch.setenv("FVB_SPATIAL_SLAB_WIDTH", "64")
    slab = hiplib.run_spatial_host(h, sp, y)
    monkeypatch.delenv("FVB_SPATIAL_SLAB_WIDTH")
    wide = hiplib.run_spatial_host(h, sp, y)
    monkeypatch.setenv("FVB_SPATIAL_PER_LEVEL", "1")
    per_level = hiplib.run_spatial_host(h, sp, y)
    for k in ("mvn", "status", "iterations"):
        assert np.array_equal(slab[k], per_level[k], equal_nan=True) and np.array_equal(wide[k], per_level[k], equal_nan=True), k


@gpu
@pytest.mark.parametrize("what", ["poly degree 4", "three exponentials", "seven regressors"])
def test_larger_parameter_counts_under_spatial_vb(what):
    """vb_spatial_more.hip: 5 - 8 parameters of the built-in models (the slab sweep with one spatial parameter among
    them), against the oracle"""
    mask, coords = masked_volume((9, 8, 6), seed=51, keep=0.9)
    V = coords.shape[1]
    rng = np.random.default_rng(52)
    smooth = 1.0 + 0.3 * np.sin(coords[0] / 2.0) + 0.2 * np.cos(coords[1] / 3.0)
    if what == "poly degree 4":
        T = 14
        t = np.arange(1, T + 1) / 4.0
        y = smooth[None, :] + 0.5 * t[:, None] - 0.2 * t[:, None] ** 2 + 0.03 * t[:, None] ** 3 + rng.normal(0, 0.05, (T, V))
        h = vbabi.build_config(vbabi.MODEL_POLY, V, T, degree=4, max_iterations=4, param_overrides={"c0": dict(type="M")})
        tol = dict(allow_floor=True)  # (monomials up to t^4)
    elif what == "three exponentials":
        T = 60
        t = np.arange(T) * 0.03
        y = smooth[None, :] * np.exp(-0.7 * t[:, None]) + 0.6 * np.exp(-3.0 * t[:, None]) + 0.3 * np.exp(-9.0 * t[:, None]) + rng.normal(0, 0.03, (T, V))
        h = vbabi.build_config(vbabi.MODEL_EXP, V, T, num_exps=3, dt=0.03, max_iterations=2, param_overrides={"amp1": dict(type="M")})
        tol = None  # chaotic like the bi-exponential fit: population
    else:
        T = 40
        tt = (np.arange(T) + 0.5) / T
        X = np.stack([np.cos(np.pi * k * tt) for k in range(7)], axis=1)
        coef = rng.uniform(-1, 1, (7, V))
        coef[0] = smooth
        y = X @ coef + rng.normal(0, 0.05, (T, V))
        h = vbabi.build_config(vbabi.MODEL_LINEAR, V, T, design=X, max_iterations=5, param_overrides={"Parameter_1": dict(type="M")})
        tol = dict()
    sp = vbabi.SpatialHolder(coords)
    cpu, cpu2, got = oracle.run_spatial(h, sp, y), oracle.run_spatial_fma(h, sp, y), hiplib.run_spatial_host(h, sp, y)
    if tol is None:
        parity.population(h, cpu, got, parity.population_stats(h, cpu, cpu2), what=what)
    else:
        parity.strict(h, cpu, got, what=what, cpu2=cpu2, **tol)
