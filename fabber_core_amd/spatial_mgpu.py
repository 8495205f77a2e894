"""Spatial VB over several GPUs: one process per GPU, the volume cut into z-slabs.

Coupling in spatial VB is local (a voxel's prior reads its first - types M, m - or first and
second - types P, p - neighbours' posterior means, priors.cc:362-370) plus two global scalars per
spatial parameter (the a_K sums, priors.cc:233-312). Voxels are ordered z slowest
(inference_vb.cc:769-793), so a z-slab is a contiguous block of the masked-voxel list and its
ghost planes (one plane each side, two for P / p) are the blocks next to it:

        local list of rank r = [ ghost planes below | owned voxels | ghost planes above ]

Per iteration (include/fabber_vb.h, "step by step"): all-reduce of the a_K sums (2 doubles per
parameter), the rank's own Gauss-Seidel sweeps, then the halo exchange of the boundary planes'
means with the two adjacent ranks. Ghost values are those of the end of the previous iteration:
exact for the slab above (the reference has not updated it yet either), one iteration old for the
slab below (block-Jacobi across the cut; the reference's strictly sequential sweep cannot run
slabs concurrently, and the per-iteration a_K reduction rules out pipelining them across
iterations). tests/test_spatial_mgpu.py measures the deviation.

torch.distributed carries the collectives ("nccl" = RCCL over xGMI in production, "gloo" in the
tests); the payloads are staged through host memory (a plane of means is ~100 KB).
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
                         ("fabber_vb_spatial_set_ak_sums", [C.c_void_p, C.c_void_p]),
                         ("fabber_vb_spatial_sweep", [C.c_void_p, C.c_int32]),
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

    def set_ak_sums(self, sums):
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        self._check(self.L.fabber_vb_spatial_set_ak_sums(self.h, sums.ctypes.data))

    def sweep(self, it):
        self._check(self.L.fabber_vb_spatial_sweep(self.h, it))

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


def _exchange(run, plan, rank, world, P, device):
    """Boundary planes to the adjacent ranks, theirs into the ghost voxels."""
    g0, b, e, g1 = plan[rank]
    on_gpu = dist.get_backend() == "nccl"
    ops, recvs = [], []

    def pack(v0, n):
        means, status = run.get(v0 - g0, n)
        t = torch.from_numpy(np.concatenate([means.ravel(), status.astype(np.float64)]))
        return t.to(device) if on_gpu else t

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
        if send_hi > send_lo:
            ops.append(dist.P2POp(dist.isend, pack(send_lo, send_hi - send_lo), nb))
        if recv_hi > recv_lo:
            n = recv_hi - recv_lo
            buf = torch.empty(n * (P + 1), dtype=torch.float64, device=device if on_gpu else "cpu")
            ops.append(dist.P2POp(dist.irecv, buf, nb))
            recvs.append((recv_lo, n, buf))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for lo, n, buf in recvs:
        a = buf.cpu().numpy()
        run.put(lo - g0, a[:n * P].reshape(P, n), a[n * P:].astype(np.int32))


def run_spatial_sharded(holder, spatial, data, device="cuda:0"):
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
        for it in range(loc.cfg.max_iterations):
            if has_spatial and (it > 0 or sp.sp.update_first_iter):
                sums = torch.from_numpy(run.ak_sums())
                if world > 1:
                    if dist.get_backend() == "nccl":
                        t = sums.to(prob.device)
                        dist.all_reduce(t, op=dist.ReduceOp.SUM)
                        sums = t.cpu()
                    else:
                        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
                run.set_ak_sums(sums.numpy())
            run.sweep(it)
            if world > 1 and has_spatial:
                _exchange(run, plan, rank, world, P, prob.device)
    finally:
        run.close()
    res = prob.results()
    lo, hi = b - g0, e - g0
    out = {k: (v[:, lo:hi] if v.ndim == 2 else v[lo:hi]) for k, v in res.items()}
    out["begin"], out["end"] = b, e
    return out
