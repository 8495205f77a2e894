"""Spatial VB over several GPUs: one process per GPU, the volume cut into z-slabs.

Coupling in spatial VB is local (a voxel's prior reads its first - types M, m - or first and
second - types P, p - neighbours' posterior means, priors.cc:362-370) plus two global scalars per
spatial parameter (the a_K sums, priors.cc:233-312). Voxels are ordered z slowest
(inference_vb.cc:769-793), so a z-slab is a contiguous block of the masked-voxel list and its
ghost planes (one plane each side, two for P / p) are the blocks next to it:

        local list of rank r = [ ghost planes below | owned voxels | ghost planes above ]

Per iteration (include/fabber_vb.h, "step by step"): all-reduce of the a_K sums (2 doubles per
parameter), the first sweep, the second sweep, then the halo exchange of the boundary planes' means
with the two adjacent ranks.

The first sweep is the reference's Gauss-Seidel sweep EXACTLY (inference_vb.cc:614-672): a voxel's
prior reads the CURRENT means of its neighbours, already updated for neighbours with a smaller index
and still the previous iteration's for the others. The single-device sweep keeps that order by
levels (level = x + y + z, or x + 2 y + 3 z when second neighbours are read: every neighbour with a
smaller index has a smaller level); the slabs keep it by running the SAME global levels as a
pipeline: the level range is cut into chunks of `chunk_levels`, slab r works on chunk c at tick
c + r and after every tick hands the means of its top boundary planes to slab r + 1. What slab r
reads from the slab below has a lower level, i.e. lies in a chunk slab r - 1 finished at least one
tick earlier (this iteration's value, as in the reference); what it reads from the slab above has a
higher level and is a ghost that nothing touches until the end-of-iteration exchange (the previous
iteration's value, as in the reference). nchunks + world - 1 ticks per iteration; the a_K
all-reduce stays once per iteration. tests/test_spatial_mgpu.py: bit-identical to the single-device
run with 2 and 3 slabs, first- and second-neighbour priors.

torch.distributed carries the messages ("nccl" = RCCL over xGMI in production: device buffers in,
device buffers out, fabber_vb_spatial_copy_means takes either; "gloo" in the tests).
"""
import copy
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import hiplib, vbabi


def halo_planes(holder):
    """Ghost planes a slab needs on each side: 2 if any prior reads second neighbours."""
    types = [holder.cfg.prior_type[k] for k in range(holder.cfg.n_params)]
    return 2 if any(t in (vbabi.PRIOR_SPATIAL_P, vbabi.PRIOR_SPATIAL_p) for t in types) else 1


def slab_plan(coords, world, halo):
    """[(g0, b, e, g1)] per rank: rank r owns global voxels [b, e) and keeps copies of [g0, b) and
    [e, g1). Cuts fall on z-plane boundaries, balanced by voxel count."""
    z = np.asarray(coords)[2]
    V = z.shape[0]
    if np.any(np.diff(z) < 0):
        raise ValueError("co-ordinates must be ordered with z slowest")
    starts = np.flatnonzero(np.diff(z, prepend=z[0] - 1))  # first voxel of every occupied plane
    cuts = []
    for r in range(1, world):
        c = int(starts[np.argmin(np.abs(starts - V * r / world))])
        cuts.append(c)
    bounds = [0] + cuts + [V]
    if any(bounds[i + 1] <= bounds[i] for i in range(world)):
        raise ValueError("too few z-planes (%d) for %d slabs" % (len(starts), world))
    plan = []
    for r in range(world):
        b, e = bounds[r], bounds[r + 1]
        g0 = int(np.searchsorted(z, z[b] - halo, side="left")) if r > 0 else b
        g1 = int(np.searchsorted(z, z[e - 1] + halo, side="right")) if r < world - 1 else e
        plan.append((g0, b, e, g1))
    for r in range(world):  # ghosts must come from the adjacent slab only
        g0, b, e, g1 = plan[r]
        if (r > 0 and g0 < plan[r - 1][1]) or (r < world - 1 and g1 > plan[r + 1][2]):
            raise ValueError("slab %d is thinner than the %d-plane halo of its neighbour" % (r, halo))
    return plan


def local_holder(holder, g0, g1):
    """A copy of the problem description restricted to voxels [g0, g1)."""
    cfg = vbabi.FvbConfig.from_buffer_copy(holder.cfg)
    keep = dict(holder.keep)
    cfg.n_voxels = g1 - g0
    for name, arr in holder.keep.items():
        if name == "init_mvn":
            keep[name] = np.ascontiguousarray(arr[:, g0:g1])
            cfg.init_mvn = keep[name].ctypes.data
        elif name.startswith("image_"):
            keep[name] = np.ascontiguousarray(arr[g0:g1])
            cfg.image_prior[int(name.split("_")[1])] = keep[name].ctypes.data
    return vbabi.ConfigHolder(cfg, holder.params, keep)


class _Run:
    """ctypes wrapper of the stepwise C entry points"""

    def __init__(self, prob, spatial, stream):
        L = hiplib.lib()
        self.L = L
        L.fabber_vb_spatial_open.restype = C.c_int32
        L.fabber_vb_spatial_open.argtypes = [C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbSpatial), C.c_void_p,
                                             C.POINTER(vbabi.FvbOutputs), C.c_void_p, C.POINTER(C.c_void_p)]
        for fn, args in (("fabber_vb_spatial_ak_sums", [C.c_void_p, C.c_void_p]),
                         ("fabber_vb_spatial_ak_segment_sums", [C.c_void_p, C.c_void_p, C.c_void_p]),
                         ("fabber_vb_spatial_set_ak_sums", [C.c_void_p, C.c_void_p]),
                         ("fabber_vb_spatial_sweep", [C.c_void_p, C.c_int32]),
                         ("fabber_vb_spatial_sweep_levels", [C.c_void_p, C.c_int32, C.c_int64, C.c_int64]),
                         ("fabber_vb_spatial_sweep_noise", [C.c_void_p, C.c_int32]),
                         ("fabber_vb_spatial_level_weights", [C.c_void_p, C.c_void_p]),
                         ("fabber_vb_spatial_fprior", [C.c_void_p, C.c_void_p, C.c_int32]),
                         ("fabber_vb_spatial_copy_means", [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
                         ("fabber_vb_spatial_close", [C.c_void_p])):
            getattr(L, fn).restype = C.c_int32
            getattr(L, fn).argtypes = args
        self.h = C.c_void_p()
        self.P = prob.cfg.n_params
        self._check(L.fabber_vb_spatial_open(C.byref(prob.cfg), C.byref(spatial.sp), prob.data.data_ptr(), C.byref(prob.out),
                                             C.c_void_p(stream.cuda_stream), C.byref(self.h)))

    def _check(self, rc):
        if rc != 0:
            raise hiplib.HipEngineError("spatial run: %d %s" % (rc, self.L.fabber_vb_last_error().decode()))

    def ak_sums(self):
        sums = np.zeros((self.P, 2))
        self._check(self.L.fabber_vb_spatial_ak_sums(self.h, sums.ctypes.data))
        return sums

    def ak_segment_sums(self):
        """[n_segments][P][2]: the a_K sums per segment of the owned voxels (a z-plane, cut every 4096 voxels)"""
        n = C.c_int32(0)
        self._check(self.L.fabber_vb_spatial_ak_segment_sums(self.h, None, C.byref(n)))
        part = np.zeros((n.value, self.P, 2))
        self._check(self.L.fabber_vb_spatial_ak_segment_sums(self.h, part.ctypes.data, C.byref(n)))
        return part

    def set_ak_sums(self, sums):
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        self._check(self.L.fabber_vb_spatial_set_ak_sums(self.h, sums.ctypes.data))

    def sweep(self, it):
        self._check(self.L.fabber_vb_spatial_sweep(self.h, it))

    def sweep_levels(self, it, lo, hi):
        self._check(self.L.fabber_vb_spatial_sweep_levels(self.h, it, int(lo), int(hi)))

    def sweep_noise(self, it):
        self._check(self.L.fabber_vb_spatial_sweep_noise(self.h, it))

    def level_weights(self):
        w = (C.c_int32 * 3)()
        self._check(self.L.fabber_vb_spatial_level_weights(self.h, w))
        return [int(x) for x in w]

    def fprior(self, value=None):
        v = C.c_double(0.0 if value is None else value)
        self._check(self.L.fabber_vb_spatial_fprior(self.h, C.byref(v), 0 if value is None else 1))
        return v.value

    # boundary planes as ONE device tensor [P * n means | n status]: no host staging
    def get_device(self, v0, n, device):
        buf = torch.empty(n * (self.P + 1), dtype=torch.float64, device=device)
        st = torch.empty(n, dtype=torch.int32, device=device)
        self._check(self.L.fabber_vb_spatial_copy_means(self.h, v0, n, buf.data_ptr(), st.data_ptr(), 0))
        buf[n * self.P:] = st.to(torch.float64)
        return buf

    def put_device(self, v0, n, buf):
        st = buf[n * self.P:].to(torch.int32).contiguous()
        self._check(self.L.fabber_vb_spatial_copy_means(self.h, v0, n, buf.data_ptr(), st.data_ptr(), 1))

    def get(self, v0, n):
        means, status = np.empty((self.P, n)), np.empty(n, dtype=np.int32)
        self._check(self.L.fabber_vb_spatial_copy_means(self.h, v0, n, means.ctypes.data, status.ctypes.data, 0))
        return means, status

    def put(self, v0, means, status):
        means, status = np.ascontiguousarray(means, dtype=np.float64), np.ascontiguousarray(status, dtype=np.int32)
        self._check(self.L.fabber_vb_spatial_copy_means(self.h, v0, status.shape[0], means.ctypes.data, status.ctypes.data, 1))

    def close(self):
        if self.h:
            h, self.h = self.h, C.c_void_p()
            self._check(self.L.fabber_vb_spatial_close(h))


def _exchange(run, plan, rank, world, P, device, directions=(-1, +1)):
    """Boundary planes to the adjacent ranks, theirs into the ghost voxels. directions: (-1, +1) both ways
    (end of an iteration); (+1,) only upwards - this rank's top planes to the slab above, the slab below's
    into this rank's ghosts - which is the hand-over between two ticks of the first sweep."""
    g0, b, e, g1 = plan[rank]
    on_gpu = dist.get_backend() == "nccl"
    ops, recvs = [], []

    def pack(v0, n):
        if on_gpu:
            return run.get_device(v0 - g0, n, device)
        means, status = run.get(v0 - g0, n)
        return torch.from_numpy(np.concatenate([means.ravel(), status.astype(np.float64)]))

    for nb in (rank - 1, rank + 1):
        if nb < 0 or nb >= world:
            continue
        ng0, nb_b, nb_e, ng1 = plan[nb]
        # what the neighbour holds as ghosts of mine / what I hold as ghosts of its
        if nb < rank:
            send_lo, send_hi = b, min(e, ng1)
            recv_lo, recv_hi = g0, b
        else:
            send_lo, send_hi = max(b, ng0), e
            recv_lo, recv_hi = e, g1
        send = (nb - rank) in directions      # my planes travel in that direction
        recv = (rank - nb) in directions      # the neighbour's planes travel towards me
        if send and send_hi > send_lo:
            ops.append(dist.P2POp(dist.isend, pack(send_lo, send_hi - send_lo), nb))
        if recv and recv_hi > recv_lo:
            n = recv_hi - recv_lo
            buf = torch.empty(n * (P + 1), dtype=torch.float64, device=device if on_gpu else "cpu")
            ops.append(dist.P2POp(dist.irecv, buf, nb))
            recvs.append((recv_lo, n, buf))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for lo, n, buf in recvs:
        if on_gpu:
            run.put_device(lo - g0, n, buf)
        else:
            a = buf.numpy()
            run.put(lo - g0, a[:n * P].reshape(P, n), a[n * P:].astype(np.int32))


def pipeline_schedule(level_min, level_max, chunk_levels, world):
    """[(tick, {rank: (lo, hi)})]: the level range [lo, hi) each slab sweeps at each tick (see the module
    docstring). Every rank appears nchunks times; rank r at tick c + r."""
    nchunks = max(1, -(-(level_max - level_min + 1) // chunk_levels))
    ticks = []
    for t in range(nchunks + world - 1):
        work = {}
        for r in range(world):
            c = t - r
            if 0 <= c < nchunks:
                work[r] = (level_min + c * chunk_levels, level_min + (c + 1) * chunk_levels)
        ticks.append((t, work))
    return ticks


def run_spatial_sharded(holder, spatial, data, device="cuda:0", chunk_levels=16):
    """Spatial VB of the WHOLE problem (holder / spatial / data describe every voxel) on this
    process' slab. Returns the results of the owned voxels and their global range:
    dict(mvn=[rows][n_owned], free_energy, status, iterations, begin, end). torch.distributed must
    be initialised when there is more than one rank."""
    from .device import DeviceProblem

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    V = holder.cfg.n_voxels
    plan = slab_plan(spatial.coords, world, halo_planes(holder))
    g0, b, e, g1 = plan[rank]
    loc = local_holder(holder, g0, g1)
    sp = vbabi.SpatialHolder(spatial.coords[:, g0:g1], spatial_dims=spatial.sp.spatial_dims, spatial_speed=spatial.sp.spatial_speed,
                             q1=spatial.sp.q1, q2=spatial.sp.q2, update_first_iter=bool(spatial.sp.update_first_iter),
                             owned=(b - g0, e - g0), n_voxels_global=V)
    prob = DeviceProblem(loc, np.ascontiguousarray(np.asarray(data)[:, g0:g1]), device)
    P = loc.cfg.n_params
    has_spatial = any(loc.cfg.prior_type[k] >= vbabi.PRIOR_SPATIAL_M for k in range(P))
    run = _Run(prob, sp, torch.cuda.current_stream(prob.device))
    try:
        ticks = None
        if world > 1:
            w = run.level_weights()
            c = np.asarray(spatial.coords, dtype=np.int64)
            level = w[0] * c[0] + w[1] * c[1] + w[2] * c[2]   # the GLOBAL level range: every rank steps through all of it
            ticks = pipeline_schedule(int(level.min()), int(level.max()), chunk_levels, world)
        for it in range(loc.cfg.max_iterations):
            if has_spatial and (it > 0 or sp.sp.update_first_iter):
                if world == 1:
                    run.set_ak_sums(run.ak_sums())
                else:
                    # The global (trace, quadratic) sums, added in the ONE order every decomposition shares: the
                    # segments of the voxel list (z-planes, cut every 4096 voxels) in voxel order. A plain
                    # all-reduce of slab totals would round differently from the single-device sum, and a_K
                    # enters every voxel's prior. (n_segments x P x 2 doubles per rank: a few KB.)
                    mine = run.ak_segment_sums()
                    parts = [None] * world
                    dist.all_gather_object(parts, mine)
                    sums = np.zeros((P, 2))
                    for part in parts:            # slabs are in z order
                        for seg in part:          # sequential, as vb_spatial_ak_reduce_kernel adds them
                            sums = sums + seg
                    run.set_ak_sums(sums)
            if world == 1:
                run.sweep(it)
                continue
            for _, work in ticks:
                if rank in work:
                    run.sweep_levels(it, *work[rank])
                # hand-over: whoever swept passes its top planes up; whoever has a slab below that swept receives
                send = rank in work and rank < world - 1
                recv = (rank - 1) in work
                if has_spatial and (send or recv):
                    _exchange_up(run, plan, rank, world, P, prob.device, send, recv)
            if loc.cfg.need_f:
                # the F term of the priors of the LAST voxel of the sweep is the last slab's (inference_vb.cc:612,689,702)
                t = torch.tensor([run.fprior() if rank == world - 1 else 0.0], dtype=torch.float64)
                if dist.get_backend() == "nccl":
                    t = t.to(prob.device)
                dist.broadcast(t, src=world - 1)
                run.fprior(float(t.cpu()[0]))
            run.sweep_noise(it)
            if has_spatial:
                _exchange(run, plan, rank, world, P, prob.device)
    finally:
        run.close()
    res = prob.results()
    lo, hi = b - g0, e - g0
    out = {k: (v[:, lo:hi] if v.ndim == 2 else v[lo:hi]) for k, v in res.items()}
    out["begin"], out["end"] = b, e
    return out


def _exchange_up(run, plan, rank, world, P, device, send, recv):
    """One hand-over of the first sweep's pipeline: this rank's top boundary planes to the slab above (if
    `send`), the slab below's into this rank's ghosts (if `recv`). Both sides of a message decide from the
    same schedule, so the sends and receives pair up."""
    g0, b, e, g1 = plan[rank]
    on_gpu = dist.get_backend() == "nccl"
    ops, got = [], None
    if send:
        ng0 = plan[rank + 1][0]
        lo, hi = max(b, ng0), e
        if on_gpu:
            payload = run.get_device(lo - g0, hi - lo, device)
        else:
            means, status = run.get(lo - g0, hi - lo)
            payload = torch.from_numpy(np.concatenate([means.ravel(), status.astype(np.float64)]))
        ops.append(dist.P2POp(dist.isend, payload, rank + 1))
    if recv and b > g0:
        n = b - g0
        buf = torch.empty(n * (P + 1), dtype=torch.float64, device=device if on_gpu else "cpu")
        ops.append(dist.P2POp(dist.irecv, buf, rank - 1))
        got = (n, buf)
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if got:
        n, buf = got
        if on_gpu:
            run.put_device(0, n, buf)
        else:
            a = buf.numpy()
            run.put(0, a[:n * P].reshape(P, n), a[n * P:].astype(np.int32))
