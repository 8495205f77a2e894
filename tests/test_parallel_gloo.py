"""The N>1 path on CPU: two gloo ranks shard a voxel list, fit their blocks, all-reduce the
summary and gather the result image. The per-rank compute is the CPU oracle here (the GPU
engine needs a device); sharding, gather order and the collective are the code bench.py and a
multi-GPU host use."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, V, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import cases
    import oracle
    from fabber_core_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    h_all, y_all = cases.exp_problem(V, 50, 1, 0.04, seed=99, max_iterations=6, need_f=True)
    b, e = parallel.shard_bounds(V, world_size, rank)
    h, _ = cases.exp_problem(e - b, 50, 1, 0.04, seed=99, max_iterations=6, need_f=True)
    res = oracle.run(h, np.ascontiguousarray(y_all[:, b:e]))
    summary = torch.tensor([res["free_energy"].sum(), float(res["iterations"].sum()), float((res["status"] != 0).sum())],
                           dtype=torch.float64)
    parallel.global_summary(summary)
    full = parallel.gather_rows(torch.from_numpy(res["mvn"]), V)
    tmax = parallel.global_max(rank + 1.5)
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), mvn=full.numpy(), summary=summary.numpy(), tmax=tmax)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("V", [301, 64])
def test_two_ranks_equal_one_process(tmp_path, V):
    sys.path.insert(0, HERE)
    import cases
    import oracle
    port = _free_port()
    mp.spawn(_worker, args=(2, port, V, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "out.npz"))
    h, y = cases.exp_problem(V, 50, 1, 0.04, seed=99, max_iterations=6, need_f=True)
    ref = oracle.run(h, y)
    assert np.array_equal(got["mvn"], ref["mvn"])  # voxels are independent: bit-identical
    assert np.isclose(got["summary"][0], ref["free_energy"].sum(), rtol=1e-12)
    assert got["summary"][1] == ref["iterations"].sum() and got["summary"][2] == 0
    assert got["tmax"] == 2.5


def test_shard_bounds_cover_and_order():
    from fabber_core_amd import parallel
    for V in (0, 1, 7, 64, 1000003):
        for ws in (1, 2, 3, 8):
            edges = [parallel.shard_bounds(V, ws, r) for r in range(ws)]
            assert edges[0][0] == 0 and edges[-1][1] == V
            assert all(edges[i][1] == edges[i + 1][0] for i in range(ws - 1))
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1
