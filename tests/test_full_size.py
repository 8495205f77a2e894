"""BASELINE configs[3] and configs[4] at their FULL sizes (256^3 voxels with AR(1) noise; 128^3 voxels with the
MRF prior), generated on the device, checked through properties that do not need a CPU run of that size:
the volume is made of replicas of an oracle-sized block, every replica must reproduce the block's result bit
for bit, and the block's result is compared with the CPU oracle."""
import numpy as np
import pytest
import torch

import cases
import oracle
import parity
from fabber_core_amd import hiplib, vbabi
from fabber_core_amd.device import DeviceProblem

pytestmark = pytest.mark.gpu


def test_c4_ar1_linear_model_at_256_cubed():
    """configs[3]: linear design (4 regressors), AR(1) noise, 200 timepoints, 256^3 = 16 777 216 voxels: 13.4 GB
    of series resident in HBM. The volume is 512 copies of a 32^3 block; the block runs against the oracle
    (strict per-voxel parity), the full volume must give every copy the block's result bit for bit."""
    nb, reps = 32 ** 3, 512
    V = nb * reps
    assert V == 256 ** 3
    hb, yb = cases.linear_problem(nb, 200, seed=20260104, max_iterations=10, noise=vbabi.NOISE_AR1)
    block = hiplib.run_host(hb, yb)
    # (over 32768 voxels the worst AR coefficient of two CPU builds is already 4e-6 apart: measured floor allowed)
    parity.strict(hb, oracle.run(hb, yb), block, what="C4 block", cpu2=oracle.run_fma(hb, yb), allow_floor=True)
    hf, _ = cases.linear_problem(8, 200, seed=20260104, max_iterations=10, noise=vbabi.NOISE_AR1)
    hf.cfg.n_voxels = V
    y_dev = torch.from_numpy(yb).to("cuda:0").repeat(1, reps).contiguous()
    assert y_dev.shape == (200, V) and y_dev.element_size() * y_dev.nelement() > 13e9
    prob = DeviceProblem(hf, y_dev, "cuda:0")
    assert prob.kernel == "lane_ar1<linear,4>"
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    prob.run()  # (first launch: allocations)
    start.record()
    prob.run()
    stop.record()
    torch.cuda.synchronize()
    print("C4 at 256^3: %.1f ms per run = %.1f M voxels/s" % (start.elapsed_time(stop), V / start.elapsed_time(stop) / 1e3))
    want = torch.from_numpy(block["mvn"]).to("cuda:0")
    got = prob.mvn.view(prob.mvn.shape[0], reps, nb)
    assert torch.equal(got, want[:, None, :].expand(-1, reps, -1))
    assert int((prob.status != 0).sum()) == 0 and int((prob.iterations != 10).sum()) == 0


def test_c5_spatial_biexponential_at_128_cubed():
    """configs[4]: bi-exponential with the 6-neighbour MRF prior on amp1 at 128^3. Voxels of a spatial run are
    coupled, so the volume is a 128^3 grid holding 4 x 4 x 4 copies of a 31^3 block separated by one-voxel gaps
    of mask: the copies have no neighbours in common and meet only in the global smoothing precision a_K, which
    is the same function of every copy. So (1) every copy must carry the same posterior bit for bit, (2) a_K of
    R copies with (q1, q2) equals a_K of ONE copy with (R q1, q2 / R) (priors.cc:310-312), and that one-block
    run fits the CPU oracle: population parity (the bi-exponential fit is chaotic per voxel)."""
    n, b, R = 128, 31, 64
    hblock, cblock, yblock, _ = cases.c5_problem((b, b, b), seed=20260105)
    # the full grid: block copies at offsets 32 i, gaps (coordinate 31, 63, ...) masked out
    offs = np.array([(i, j, k) for k in range(4) for j in range(4) for i in range(4)]) * 32
    coords = np.concatenate([cblock + o[:, None] for o in offs], axis=1)
    order = np.lexsort((coords[0], coords[1], coords[2]))  # voxel order of a masked volume: x fastest, z slowest
    coords = np.ascontiguousarray(coords[:, order]).astype(np.int32)
    Vb = cblock.shape[1]
    V = R * Vb
    src = np.tile(np.arange(Vb), R)[order]        # which block voxel a voxel of the volume copies
    copy = np.repeat(np.arange(R), Vb)[order]
    assert coords.max() < n and V == 1906624
    y_dev = torch.from_numpy(yblock).to("cuda:0")[:, torch.from_numpy(src).to("cuda:0")].contiguous()
    hf, _, _, _ = cases.c5_problem((2, 2, 2))
    hf.cfg.n_voxels = V
    prob = DeviceProblem(hf, y_dev, "cuda:0")
    prob.run_spatial(vbabi.SpatialHolder(coords))
    mvn = prob.mvn.cpu().numpy()
    status = prob.status.cpu().numpy()
    first = mvn[:, copy == 0]
    for r in range(1, R):
        assert np.array_equal(mvn[:, copy == r], first, equal_nan=True), r
    # one block with the rescaled hyper-prior against the oracle
    sp1 = vbabi.SpatialHolder(cblock, q1=10.0 * R, q2=1.0 / R)
    cpu, cpu2 = oracle.run_spatial(hblock, sp1, yblock), oracle.run_spatial_fma(hblock, sp1, yblock)
    got = dict(mvn=first, status=status[copy == 0] & 0xFF, iterations=np.full(Vb, 10, dtype=np.int32))
    s = parity.population(hblock, cpu, got, parity.population_stats(hblock, cpu, cpu2), what="C5 block of the full volume")
    print("C5 at 128^3 (64 copies of 31^3): copies identical; block vs oracle", s)
