"""Multi-GPU plumbing: one process per GPU, voxels sharded in contiguous blocks.

Voxelwise VB has no data-path exchange between voxels (inference_vb.cc:423-571 touches only
voxel v's state), so rank r simply fits block r of the masked-voxel list; the only collective is
a tiny all-reduce of a per-rank summary [sum F (or checksum), sum iterations, bad voxels] - the
global convergence / health report. Backend "nccl" is RCCL on ROCm; the same code runs on "gloo"
for the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_voxels, world_size, rank):
    """[begin, end) of rank's contiguous block; the first n_voxels % world_size ranks get one
    voxel more. Concatenating the blocks in rank order restores the caller's voxel order."""
    base, rem = divmod(int(n_voxels), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def global_summary(local):
    """Sum a small per-rank tensor over all ranks (in place) and return it."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(local, op=dist.ReduceOp.SUM)
    return local


def global_max(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local, n_voxels_total):
    """Gather a [rows][local_voxels] tensor from every rank into [rows][n_voxels_total] on every
    rank (blocks may differ in size by one voxel)."""
    ws, rank = world()
    if ws == 1:
        return local
    rows = local.shape[0]
    width = max(shard_bounds(n_voxels_total, ws, r)[1] - shard_bounds(n_voxels_total, ws, r)[0] for r in range(ws))
    padded = torch.zeros((rows, width), dtype=local.dtype, device=local.device)
    padded[:, :local.shape[1]] = local
    parts = [torch.empty_like(padded) for _ in range(ws)]
    dist.all_gather(parts, padded)
    out = torch.empty((rows, n_voxels_total), dtype=local.dtype, device=local.device)
    for r in range(ws):
        b, e = shard_bounds(n_voxels_total, ws, r)
        out[:, b:e] = parts[r][:, :e - b]
    return out
