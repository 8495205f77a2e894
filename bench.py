#!/usr/bin/env python3
"""Headline benchmark: voxels/s of the voxelwise VB hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c1|c4|c5] [--voxels V | --grid N]

A "step" is one complete pass of the hot path (every voxel of this rank's shard fitted to VB
convergence: fabber_vb_run_device = Vb::DoCalculationsVoxelwise) over one batch of synthetic
input that is already resident in HBM. Default workload = BASELINE.json configs[2], the
configuration the north star's target is quoted on and which fits one GPU: bi-exponential model,
white noise, 100 timepoints, max-iterations 50, 1e6 voxels PER GPU (weak scaling: N ranks fit
N x 1e6 voxels; voxels are independent, so the shard is a contiguous block and the only
collective is one tiny all-reduce of [sum F or noise checksum, sum iterations, bad voxels]).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - HBM roofline of the VB kernel: algorithmic bytes (4 T in + 4 rows out per voxel,
                 SURVEY.md section 8d) / HIP-event kernel time, against 8 TB/s; `traffic` = HBM bytes
                 per launch from the PMC passes committed under profiles/ (tools/pmc_passes.sh). The
                 kernel is bound by fp64 VALU issue, not by HBM: roofline.valu_issue gives the executed
                 instruction count and issue utilisation from the same profile.
  cpu_baseline - the CPU oracle (port of the reference algorithm, single thread) timed on this
                 host on a bounded voxel sample, plus the max relative difference of the
                 posterior means between GPU and CPU on that sample.
  e2e          - (N = 1) the boundary itself with PCIe and host work included - fabber_vb_run_host through host pointers and
                 the reference's C ABI stage by stage - measured in a CHILD process without PyTorch (--e2e-child): the
                 system's HIP runtime, as a caller of the C ABI has it. Never the headline `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# --e2e-child: this process stands for a caller of the C ABI (the `e2e` object is measured in it): no torch, the
# system's HIP runtime - see bench_boundary_in_child
E2E_CHILD = "--e2e-child" in sys.argv
if E2E_CHILD:
    os.environ["FVB_NO_TORCH"] = "1"

import numpy as np  # noqa: E402
if not E2E_CHILD:
    import torch  # noqa: E402
    import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (model kwargs, T, default voxels per GPU, iterations, description)
    "c3": dict(num_exps=2, T=100, dt=0.02, voxels=1_000_000, its=50,
               desc="BASELINE configs[2]: bi-exponential (examples biexp), white noise, 100 timepoints, max-iterations 50"),
    "c2": dict(num_exps=1, T=50, dt=0.04, voxels=128 * 128 * 64, its=10,
               desc="BASELINE configs[1]: exp single-exponential, white noise, 50 timepoints, 128x128x64 voxels, max-iterations 10"),
    "c1": dict(kind="poly", degree=2, T=10, voxels=8 * 8 * 8, its=10,
               desc="BASELINE configs[0]: poly degree 2, white noise, 10 timepoints, 8x8x8 volume, max-iterations 10"),
    "c5": dict(kind="spatial", num_exps=2, T=100, dt=0.02, grid=128, voxels=128 ** 3, its=10,
               desc="BASELINE configs[4]: spatial VB, bi-exponential with a 6-neighbour MRF prior (type M) on amp1, "
                    "128^3 voxels, 100 timepoints, max-iterations 10 (one GPU: the whole volume)"),
    "c4": dict(kind="linear_ar", T=200, voxels=2_000_000, its=10,
               desc="BASELINE configs[3] model: linear design (4 regressors), AR(1) noise, 200 timepoints, max-iterations 10; "
                    "2e6 voxels per GPU by default (the 256^3 volume is --voxels 16777216: 13 GB of series)"),
}


def usable_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def make_problem(w, V, seed, need_f):
    import cases
    from fabber_core_amd import vbabi
    kind = w.get("kind", "exp")
    if kind == "exp":
        return cases.exp_problem(V, w["T"], w["num_exps"], w["dt"], seed=seed, max_iterations=w["its"], need_f=need_f)
    if kind == "poly":
        return cases.poly_problem(V, w["T"], w["degree"], seed=seed, max_iterations=w["its"], need_f=need_f)
    return cases.linear_problem(V, w["T"], seed=seed, max_iterations=w["its"], need_f=need_f, noise=vbabi.NOISE_AR1)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz (= half the 157.3 TF fp32 vector peak)
N_SIMDS = 256 * 4              # MI355X: 256 CUs x 4 SIMDs


def pmc_profile(workload, V, kernel, args, kernel_ms):
    """What the PMC passes committed under profiles/ say about the VB kernel (tools/pmc_passes.sh: FETCH_SIZE and
    WRITE_SIZE in separate rocprofv3 --pmc passes of this very command, scaled by the bytes-per-count measured with
    tools/pmc_calib on the kernels' own access widths; the fp64 instruction counters and GRBM_GUI_ACTIVE in a third).
    Counters cannot be read from inside the timed run, so the figures are attached only when the committed profile
    is of the same workload, size, kernel and default engine settings. Every rate and fraction here is the PROFILE's:
    a pass's counters over the kernel's duration in that same pass (the live kernel time of this run is roofline.kernel_ms;
    it is not mixed into them)."""
    path = os.path.join(ROOT, "profiles", "r4_pmc_%s%s.json" % (workload, "_needf" if args.need_f else ""))
    if not os.path.exists(path) or V != WORKLOADS[workload]["voxels"] or args.variant != "auto" \
            or args.residual != "auto" or args.residual_tol is not None:
        return {"traffic": None}
    prof = json.load(open(path))
    name = (prof.get("kernel_trace") or {}).get("name", "")
    tag = kernel[kernel.find("<") + 1:].rstrip(">").split(",")  # e.g. lane<exp,4> / lane_ar1<linear,4> -> ["exp", "4"]
    if len(tag) < 2 or ("%sModel<%s>" % (tag[0].capitalize(), tag[1])) not in name:
        return {"traffic": None}
    cal, ctr, der = prof["calibration"], prof["counters"], prof.get("derived", {})
    fetch = ctr["FETCH_SIZE"]["mean_per_launch"] * cal["read_rows<float> FETCH_SIZE"]["bytes_per_count"]
    write = ctr["WRITE_SIZE"]["mean_per_launch"] * cal["write_rows<double> WRITE_SIZE"]["bytes_per_count"]
    out = {"traffic": fetch + write,
           "traffic_detail": {"fetch_bytes": fetch, "write_bytes": write, "source": os.path.relpath(path, ROOT),
                              "profiled_kernel_ms": prof["kernel_trace"]["avg_ns"] / 1e6,
                              "note": "the series is re-read on every pass (51 passes; at this size it does not stay in L2), "
                                      "the tiled copy is written and read once more, plus what is left of register spills"}}
    clock = der.get("shader_clock_ghz")
    valu = ctr.get("SQ_INSTS_VALU", {}).get("mean_per_launch")
    if valu and der.get("valu_issue_utilisation"):
        # one fp64 VALU instruction of a 64-lane wave occupies its SIMD's 16-lane pipe for 4 cycles
        out["valu_issue"] = {"valu_wave_instructions": valu, "simds": N_SIMDS, "shader_clock_ghz_measured": clock,
                             "cycles_source": "GRBM_GUI_ACTIVE / 8 of the same counter pass as SQ_INSTS_VALU",
                             "utilisation": der["valu_issue_utilisation"],
                             "wait_any_over_wave_cycles": ctr["SQ_WAIT_ANY"]["mean_per_launch"] / ctr["SQ_WAVE_CYCLES"]["mean_per_launch"]}
    if der.get("fp64_flop_per_launch"):
        out["fp64_executed"] = {"flop_per_launch": der["fp64_flop_per_launch"], "achieved": der["fp64_tflops"], "unit": "TFLOP/s",
                                "vector_peak_at_2p4ghz": FP64_VALU_PEAK_TFLOPS, "frac": der["fp64_frac_of_peak"],
                                "profiled_kernel_ms": der["fp64_pass_avg_ns"] / 1e6,
                                "counted": "(2 x SQ_INSTS_VALU_FMA_F64 + ADD_F64 + MUL_F64) x 64 lanes, executed, over the "
                                           "kernel's duration in the pass that counted them"}
    return out


def bench_spatial(args, world, rank, device):
    """--workload c5: one step = one complete spatial VB run (Vb::DoCalculationsSpatial: geometry, set-up and
    `its` Gauss-Seidel sweeps) of the whole volume on one GPU, series resident in HBM."""
    if world != 1:
        return bench_spatial_multi(args, world, rank, device)
    import cases
    from fabber_core_amd import vbabi
    from fabber_core_amd.device import DeviceProblem
    w = WORKLOADS["c5"]
    n = args.grid or w["grid"]
    extra = {} if args.prior_type == "M" else {"param_overrides": {"amp1": dict(type=args.prior_type)}}
    holder, coords, y, _ = cases.c5_problem((n, n, n), max_iterations=w["its"], need_f=bool(args.need_f), **extra)
    V, T, P = holder.cfg.n_voxels, w["T"], holder.cfg.n_params
    sp = vbabi.SpatialHolder(coords)
    prob = DeviceProblem(holder, y, device)
    for _ in range(args.warmup):
        prob.run_spatial(sp)
    torch.cuda.synchronize(device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        prob.run_spatial(sp)  # returns when the stream has drained
        ev[i][1].record()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    dev_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    got = prob.results()
    rows = holder.n_mvn_rows
    alg_bytes = (4 * T + 4 * rows + w["its"] * 6 * 8) * V  # C3's bytes + 6 neighbour means per iteration
    roofline = {"bound": "hbm", "kernel": "spatial<exp,%d>: vb_spatial_* kernels of one run" % P,
                "achieved": alg_bytes / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": dev_ms,
                "note": "device time of one run between HIP events (all its kernels; the host builds no tables in between); "
                        "per-kernel durations: profiles/r4_kernel_stats_c5.csv"}
    # HBM bytes of one run from the committed counter passes of this very command (tools/r4_profiles.sh ->
    # tools/pmc_c5_merge.py): attached when the profile is of the same problem
    pmc = os.path.join(ROOT, "profiles", "r4_pmc_c5.json")
    if os.path.exists(pmc) and n == w["grid"] and not args.need_f and args.prior_type == "M":
        prof = json.load(open(pmc))
        roofline["traffic"] = prof["per_run"]["traffic"]
        roofline["traffic_detail"] = {"source": "profiles/r4_pmc_c5.json", "fetch_bytes": prof["per_run"]["fetch_bytes"],
                                      "write_bytes": prof["per_run"]["write_bytes"],
                                      "profiled_kernel_ms": prof["per_run"]["kernel_ms"],
                                      "by_kernel_gb": {k: (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_per_run"] / 1e9
                                                       for k, v in prof["kernels"].items()},
                                      "note": prof["per_run"]["note"]}
    cpu = None
    if args.cpu_sample > 0:
        import oracle
        import parity
        m = min(32, n)
        keep = np.flatnonzero((coords[0] < m) & (coords[1] < m) & (coords[2] < m))
        hs, _, _, _ = cases.c5_problem((m, m, m), max_iterations=w["its"], need_f=bool(args.need_f), **extra)  # (config only)
        ys = np.ascontiguousarray(y[:, keep])
        sps = vbabi.SpatialHolder(np.ascontiguousarray(coords[:, keep]))
        c0 = time.perf_counter()
        ref = oracle.run_spatial(hs, sps, ys)
        cpu_s = time.perf_counter() - c0
        from fabber_core_amd import hiplib
        sub = hiplib.run_spatial_host(hs, sps, ys)
        floor = parity.population_stats(hs, ref, oracle.run_spatial_fma(hs, sps, ys))
        stats = parity.population_stats(hs, ref, sub)
        tolist = lambda d: {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in d.items()}
        cpu = {"value": len(keep) / cpu_s, "unit": "voxels/s", "cores": 1, "kind": "port",
               "sample": "the %d^3 corner block of the volume as a spatial problem of its own, same model / prior / iterations, "
                         "oracle/liboracle.so single thread, %.1f s" % (m, cpu_s),
               "gpu_over_cpu_single_thread": (V / (dev_ms * 1e-3)) / (len(keep) / cpu_s),
               "gpu_vs_cpu_on_the_block": tolist(stats), "cpu_vs_cpu_fma_build_floor": tolist(floor)}
    result = {"metric": "voxels/sec to VB convergence", "value": V * args.steps / elapsed, "unit": "voxels/s", "n_gpus": 1,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "config": {"workload": w["desc"].replace("(type M)", "(type %s)" % args.prior_type), "grid": [n, n, n], "total_voxels": V, "timepoints": T, "params": P,
                         "iterations": w["its"], "need_f": bool(args.need_f), "input_dtype": "f32", "spatial_prior_type": args.prior_type,
                         "bad_voxels": int(np.count_nonzero(got["status"])), "ms_per_iteration": dev_ms / w["its"]},
              "roofline": roofline, "cpu_baseline": cpu}
    print(json.dumps(result), flush=True)
    return result


def bench_spatial_multi(args, world, rank, device):
    """--workload c5 --gpus N: the ONE volume of config 5 cut into N z-slabs, one per GPU (strong scaling). The slabs
    sweep together - the lowest plane of slab r + 1 waits, voxel by voxel, for the means slab r's highest plane writes
    into its inboxes through peer memory (vb_spatial.h) - so they are driven by ONE process: rank 0, through
    fabber_vb_spatial_multi_* with every slab's series resident on its device before the timed region; the other ranks
    of the launcher only keep the barriers. One step = one complete run (geometry, set-up, iterations, result images
    packed on the devices), as at N = 1. FVB_BENCH_REHEARSAL=1: all slabs on the one GPU of a one-GPU box."""
    import cases
    from fabber_core_amd import hiplib, vbabi
    w = WORKLOADS["c5"]
    n = args.grid or w["grid"]
    rehearsal = os.environ.get("FVB_BENCH_REHEARSAL") == "1"
    launched = dist.is_available() and dist.is_initialized()

    def barrier():
        if launched:
            dist.barrier()
        torch.cuda.synchronize(device)

    if rank != 0:
        barrier()
        barrier()
        return None
    extra = {} if args.prior_type == "M" else {"param_overrides": {"amp1": dict(type=args.prior_type)}}
    holder, coords, y, _ = cases.c5_problem((n, n, n), max_iterations=w["its"], need_f=bool(args.need_f), **extra)
    V, T, P = holder.cfg.n_voxels, w["T"], holder.cfg.n_params
    sp = vbabi.SpatialHolder(coords)
    devices = [0] * world if rehearsal else list(range(world))
    run = hiplib.SpatialMultiRun(holder, sp, y, devices, want=("free_energy", "status", "iterations") if args.need_f else ("status", "iterations"))
    for _ in range(args.warmup):
        run.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.run()  # returns when every slab's stream has drained
    barrier()
    elapsed = time.perf_counter() - t0
    n_slabs, route = run.slabs()
    got = run.results()
    run.close()
    rows = holder.n_mvn_rows
    alg_bytes = (4 * T + 4 * rows + w["its"] * 6 * 8) * V
    ms = elapsed / args.steps * 1e3
    result = {"metric": "voxels/sec to VB convergence", "value": V * args.steps / elapsed, "unit": "voxels/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
              "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "config": {"workload": w["desc"].replace("(type M)", "(type %s)" % args.prior_type), "grid": [n, n, n], "total_voxels": V, "timepoints": T, "params": P,
                         "iterations": w["its"], "need_f": bool(args.need_f), "input_dtype": "f32", "spatial_prior_type": args.prior_type,
                         "slabs": n_slabs, "route": route, "bad_voxels": int(np.count_nonzero(got["status"])),
                         "parallelism": "%d z-slab(s) on %s, driven by rank 0 through fabber_vb_spatial_multi_* (series resident per device; "
                                        "inboxes between slabs through peer memory)" % (n_slabs, "ONE GPU (rehearsal: the numbers say nothing about "
                                        "scaling)" if rehearsal else "%d GPUs" % world)},
              "roofline": {"bound": "hbm", "kernel": "spatial<exp,%d>: vb_spatial_* kernels of one run on every slab" % P,
                           "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * (1 if rehearsal else world), "unit": "GB/s",
                           "frac": alg_bytes / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * (1 if rehearsal else world)), "traffic": None,
                           "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": ms,
                           "note": "wall time of one run on the host clock between barriers (several devices: no single stream to put events on)"},
              "cpu_baseline": None}
    print(json.dumps(result), flush=True)
    return result


def bench_boundary(w, V, holder, y, need_f, steps=3):
    """The drop-in boundary itself, PCIe and host work included (never the headline `value`):
    (1) fabber_vb_run_host - host pointers in and out of the engine's own entry point (pageable buffers the caller
        allocated beforehand): pipelined over voxel blocks, and as one block (round 2's path) beside it;
    (2) the reference's C ABI on the same problem - fabber_new, fabber_set_extent, fabber_set_opt, fabber_set_data
        (float32 [t][z][y][x] -> NEWMAT-style double matrix), fabber_dorun (Vb::DoCalculations through
        fabber_vb_run_host with the series as float64, then SaveResults), fabber_get_data of finalMVN and the means
        (fabber_capi.cc:96-233, rundata_array.cc:68-133), timed stage by stage."""
    import ctypes as C
    from fabber_core_amd import fabber, hiplib
    out = {}
    res = hiplib.run_host(holder, y)

    def timed(fn, n=steps):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.min(ts)), float(np.mean(ts))
    prev = os.environ.get("FVB_HOST_BLOCK_VOXELS")
    try:
        os.environ.pop("FVB_HOST_BLOCK_VOXELS", None)
        best, mean = timed(lambda: hiplib.run_host(holder, y, into=res))
        out["fabber_vb_run_host_ms"] = {"min": best, "mean": mean, "voxels_per_s": V / (mean * 1e-3),
                                        "how": "blocks of 262144 voxels (half-size at both ends): upload, fit (two streams) and download overlap; streams and block buffers kept between calls"}
        os.environ["FVB_HOST_BLOCK_VOXELS"] = "0"
        best, mean = timed(lambda: hiplib.run_host(holder, y, into=res))
        out["fabber_vb_run_host_one_block_ms"] = {"min": best, "mean": mean}
        os.environ.pop("FVB_HOST_BLOCK_VOXELS", None)
        # the caller keeps its buffers and has pinned them (fabber_vb_pin_host_buffer): asynchronous DMA both ways
        L = hiplib.lib()
        L.fabber_vb_pin_host_buffer.argtypes = [C.c_void_p, C.c_uint64]
        L.fabber_vb_unpin_host_buffer.argtypes = [C.c_void_p]
        pinned = []
        t0 = time.perf_counter()
        for a in [y] + [res[k] for k in ("mvn", "free_energy", "iterations", "f_history_len") if k in res]:
            if L.fabber_vb_pin_host_buffer(a.ctypes.data, a.nbytes) == 0:
                pinned.append(a)
        pin_ms = (time.perf_counter() - t0) * 1e3
        try:
            hiplib.run_host(holder, y, into=res)
            best, mean = timed(lambda: hiplib.run_host(holder, y, into=res))
            out["fabber_vb_run_host_pinned_buffers_ms"] = {"min": best, "mean": mean, "voxels_per_s": V / (mean * 1e-3),
                                                           "pinning_once_ms": pin_ms, "buffers_pinned": len(pinned)}
        finally:
            for a in pinned:
                L.fabber_vb_unpin_host_buffer(a.ctypes.data)
    finally:
        if prev is None:
            os.environ.pop("FVB_HOST_BLOCK_VOXELS", None)
        else:
            os.environ["FVB_HOST_BLOCK_VOXELS"] = prev
    if w.get("kind", "exp") != "exp":
        return out
    # (2) the reference's C ABI; the volume is V x 1 x 1 (x fastest = the voxel order of the series as it is)
    L = fabber.load_library()
    err = C.create_string_buffer(255)
    mask = np.ones(V, dtype=np.int32)
    opts = {"model": "exp", "num-exps": w["num_exps"], "dt": w["dt"], "max-iterations": w["its"], "noise": "white", "method": "vb",
            "save-mean": "", "save-mvn": "", "allow-bad-voxels": ""}
    if need_f:
        opts["save-free-energy"] = ""
    rows = holder.n_mvn_rows
    mvn = np.empty((rows, V), dtype=np.float32)
    mean = np.empty(V, dtype=np.float32)
    stages = {k: [] for k in ("new+extent+options", "set_data", "dorun", "get_data", "destroy", "total")}
    log = C.create_string_buffer(1 << 16)
    # the first run of a process also pays for what stays afterwards (the device's first use by this library, the host
    # library's image buffers - kept between handles - and their first touch): reported apart as "first_run_total"
    for i in range(steps + 1):
        t = [time.perf_counter()]
        fab = L.fabber_new(err)
        assert L.fabber_set_extent(fab, V, 1, 1, mask.ctypes.data, err) == 0, err.value
        for k, v in opts.items():
            assert L.fabber_set_opt(fab, k.encode(), str(v).encode(), err) == 0, err.value
        t.append(time.perf_counter())
        assert L.fabber_set_data(fab, b"data", w["T"], y.ctypes.data, err) == 0, err.value
        t.append(time.perf_counter())
        assert L.fabber_dorun(fab, len(log), log, err, None) == 0, err.value
        t.append(time.perf_counter())
        assert L.fabber_get_data_size(fab, b"finalMVN", err) == rows, err.value
        assert L.fabber_get_data(fab, b"finalMVN", mvn.ctypes.data, err) == 0, err.value
        for name in ("mean_amp1", "mean_r1"):
            assert L.fabber_get_data(fab, name.encode(), mean.ctypes.data, err) == 0, err.value
        t.append(time.perf_counter())
        L.fabber_destroy(fab)
        t.append(time.perf_counter())
        for k, a, b in zip(list(stages)[:5], t[:-1], t[1:]):
            stages[k].append((b - a) * 1e3)
        stages["total"].append((t[-1] - t[0]) * 1e3)
    out["fabber_capi_ms"] = {k: float(np.mean(v[1:])) for k, v in stages.items()}
    out["fabber_capi_ms"]["first_run_total"] = float(stages["total"][0])
    out["fabber_capi_ms"]["voxels_per_s"] = V / (out["fabber_capi_ms"]["total"] * 1e-3)
    out["fabber_capi_ms"]["how"] = ("fabber_new .. fabber_destroy per run, mean of %d runs after the process's first; the C ABI takes float32 "
                                    "volumes and returns float32 volumes" % steps)
    return out


def bench_boundary_in_child(args, V):
    """The `e2e` object, measured in a CHILD process that loads nothing but the C libraries - what a caller of the C ABI
    is. In this script's own process PyTorch has initialised the GPU first, and the same fabber_vb_run_host call then
    takes ~2 ms longer (20.7 - 21.2 against 18.5 - 18.9 ms on C3: tools/measure/e2e_order.py; which of torch's settings
    does it was not pursued). The child builds the same seeded problem; this process idles meanwhile."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--e2e-child", "--workload", args.workload, "--voxels", str(V)]
    if args.need_f:
        cmd.append("--need-f")
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            out = json.loads(lines[-1])
            out["measured_in"] = ("a child process without PyTorch (the system's HIP runtime, the C libraries only): what a caller of "
                                  "the C ABI sees")
            return out
        why = "exit code %d: %s" % (p.returncode, p.stderr.strip()[-300:])
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        why = repr(e)
    return {"error": "the e2e child did not deliver (%s)" % why}


def bench_single_process(args):
    """--single-process: N devices driven by ONE process through the C ABI's own sharding
    (fabber_vb_run_host_multi: contiguous voxel blocks, one host thread + stream per device, nothing
    exchanged). Host buffers in, host buffers out, so the figure INCLUDES the PCIe transfers: it is the
    rate a fabber_dorun caller with devices=all sees, not the device-resident headline."""
    from fabber_core_amd import hiplib
    w = WORKLOADS[args.workload]
    n = args.gpus
    visible = hiplib.device_count()
    if visible < n:
        raise SystemExit("--gpus %d but only %d device(s) visible" % (n, visible))
    if w.get("kind") == "spatial":
        # C5 on N devices of the node: z-slabs of the ONE volume (strong scaling), pipelined first sweep, boundary
        # planes device to device (fabber_vb_run_spatial_host_multi); host buffers in and out
        import cases
        from fabber_core_amd import vbabi
        g = args.grid or w["grid"]
        holder, coords, y, _ = cases.c5_problem((g, g, g), max_iterations=w["its"], need_f=bool(args.need_f))
        sp = vbabi.SpatialHolder(coords)
        V = holder.cfg.n_voxels
        devices = list(range(n))
        for _ in range(args.warmup):
            hiplib.run_spatial_host(holder, sp, y, devices=devices)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = hiplib.run_spatial_host(holder, sp, y, devices=devices)
        elapsed = time.perf_counter() - t0
        result = {"metric": "voxels/sec to VB convergence", "value": V * args.steps / elapsed, "unit": "voxels/s", "n_gpus": n,
                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                  "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                  "config": {"workload": w["desc"], "total_voxels": V, "timepoints": w["T"], "params": holder.cfg.n_params,
                             "iterations": w["its"], "need_f": bool(args.need_f),
                             "parallelism": "one process, fabber_vb_run_spatial_host_multi: %d z-slab(s), host buffers (PCIe-inclusive)" % n,
                             "input_dtype": "f32", "bad_voxels": int(np.count_nonzero(res["status"]))},
                  "roofline": None, "cpu_baseline": None}
        print(json.dumps(result), flush=True)
        return result
    V = (args.voxels or w["voxels"]) * n
    holder, y = make_problem(w, V, 20260103, bool(args.need_f))
    devices = list(range(n))
    # (the result arrays are the caller's, allocated once: what is timed is the engine's call - upload, fit, download)
    res = hiplib.run_host(holder, y, devices=devices)
    for _ in range(args.warmup):
        hiplib.run_host(holder, y, devices=devices, into=res)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = hiplib.run_host(holder, y, devices=devices, into=res)
    elapsed = time.perf_counter() - t0
    s_f, s_it, bad = res["summary"]
    result = {"metric": "voxels/sec to VB convergence", "value": V * args.steps / elapsed, "unit": "voxels/s", "n_gpus": n,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "config": {"workload": w["desc"], "voxels_per_gpu": V // n, "total_voxels": V, "timepoints": w["T"],
                         "params": holder.cfg.n_params, "iterations": w["its"], "need_f": bool(args.need_f),
                         "parallelism": "one process, fabber_vb_run_host_multi over %d device(s), host buffers (PCIe-inclusive)" % n,
                         "input_dtype": "f32", "mean_iterations": s_it / V, "bad_voxels": int(bad)},
              "roofline": None, "cpu_baseline": None}
    print(json.dumps(result), flush=True)
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--voxels", type=int, default=None, help="voxels per GPU (default: the workload's)")
    ap.add_argument("--grid", type=int, default=None, help="c5: side of the cubic volume (default 128)")
    ap.add_argument("--prior-type", default="M", choices=["M", "m", "P", "p"], help="c5: the spatial prior on amp1 (BASELINE config 5: M)")
    ap.add_argument("--need-f", action="store_true", help="also evaluate the free energy 4x per iteration (CLI default of the reference)")
    ap.add_argument("--cpu-sample", type=int, default=32768, help="voxels timed on the CPU oracle (0 = skip)")
    ap.add_argument("--variant", default="auto", choices=["auto", "lane", "wave"])
    ap.add_argument("--residual", default="auto", choices=["auto", "exact", "moments"],
                    help="how k'Qk is obtained (fabber_vb_set_residual_mode); default adaptive")
    ap.add_argument("--residual-tol", type=float, default=None)
    ap.add_argument("--no-e2e", action="store_true", help="skip the timing of the host-pointer entry point and of the C ABI (the `e2e` object)")
    ap.add_argument("--single-process", action="store_true",
                    help="with --gpus N: one process, the C++ engine's own sharding (fabber_vb_run_host_multi, host buffers: "
                         "the rate includes the PCIe transfers and is reported as such)")
    ap.add_argument("--e2e-child", action="store_true", help="(internal) measure the `e2e` object in this process and print it")
    args = ap.parse_args()

    if args.e2e_child:
        w = WORKLOADS[args.workload]
        V = args.voxels or w["voxels"]
        holder, y = make_problem(w, V, 20260103, bool(args.need_f))
        print(json.dumps(bench_boundary(w, V, holder, y, bool(args.need_f))), flush=True)
        return None
    if args.single_process:
        return bench_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher: start the ranks here, BEFORE anything in this process touches the GPU
        # (a process that has initialised HIP must not exec another program), and relay rank 0's line.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        child = subprocess.run(cmd, env=env)
        raise SystemExit(child.returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the VB engine has no CPU fallback")
    # FVB_BENCH_REHEARSAL=1: several ranks share ONE GPU and talk over gloo - a rehearsal of the
    # multi-rank code path on a box with a single GPU (the numbers mean nothing and the line says so)
    rehearsal = os.environ.get("FVB_BENCH_REHEARSAL") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # "nccl" is RCCL on ROCm

    import cases
    from fabber_core_amd import hiplib, parallel
    from fabber_core_amd.device import DeviceProblem

    if WORKLOADS[args.workload].get("kind") == "spatial":
        return bench_spatial(args, world, rank, device)
    hiplib.set_variant(args.variant)
    hiplib.set_residual_mode(args.residual)
    if args.residual_tol is not None:
        hiplib.set_residual_tolerance(args.residual_tol)
    w = WORKLOADS[args.workload]
    V = args.voxels or w["voxels"]
    T = w["T"]
    # every rank gets its own contiguous block of the (world x V)-voxel problem: distinct seed
    holder, y = make_problem(w, V, 20260103 + rank, bool(args.need_f))
    P = holder.cfg.n_params
    prob = DeviceProblem(holder, y, device)
    summary = torch.zeros(3, dtype=torch.float64, device=device)
    n_mvn = P + holder.n_noise_outputs
    noise_row = n_mvn * (n_mvn + 1) // 2 + P  # mean of the first noise parameter

    def step():
        prob.run()
        # per-step global health/convergence summary: [sum F (or noise-mean checksum), sum iterations, bad voxels]
        summary[0] = prob.free_energy.sum() if args.need_f else prob.mvn[noise_row].sum()
        summary[1] = prob.iterations.sum(dtype=torch.float64)
        summary[2] = (prob.status != 0).sum(dtype=torch.float64)
        parallel.global_summary(summary)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()      # same stream the kernel is launched on (torch's current stream)
        prob.run()
        ev[i][1].record()
        summary[0] = prob.free_energy.sum() if args.need_f else prob.mvn[noise_row].sum()
        summary[1] = prob.iterations.sum(dtype=torch.float64)
        summary[2] = (prob.status != 0).sum(dtype=torch.float64)
        parallel.global_summary(summary)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    elapsed = parallel.global_max(elapsed, device)
    summ = summary.cpu().numpy()

    result = None
    if rank == 0:
        total_voxels = V * world
        value = total_voxels * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        rows = holder.n_mvn_rows
        alg_bytes = (4 * T + 4 * rows) * V
        mean_its = summ[1] / (V * world)
        roofline = {
            "bound": "hbm", "kernel": prob.kernel,
            "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_ms,
            "note": "the kernel is bound by fp64 VALU issue, not by HBM: see valu_issue (executed instructions and the shader "
                    "clock measured during the kernel, from the committed PMC profile) and fp64_executed (executed fp64 "
                    "FLOP/s against the 78.6 TFLOP/s vector peak)",
        }
        roofline.update(pmc_profile(args.workload, V, prob.kernel, args, k_ms))
        cpu = None
        if args.cpu_sample > 0 and world == 1:  # the CPU baseline is timed on rank 0 of the 1-GPU run only
            import oracle
            import parity
            ns = min(args.cpu_sample, V)
            hs, _ = make_problem(w, ns, 1, bool(args.need_f))
            ys = np.ascontiguousarray(y[:, :ns])
            oracle.run(hs, ys, v_end=min(64, ns))  # page the library in
            c0 = time.perf_counter()
            ref = oracle.run(hs, ys)
            cpu_s = time.perf_counter() - c0
            # the same sample on every host core the process may use (threads over voxel ranges;
            # the oracle keeps no state between voxels and ctypes releases the GIL)
            from concurrent.futures import ThreadPoolExecutor
            ncores = usable_cores()
            edges = np.linspace(0, ns, ncores + 1).astype(int)
            c0 = time.perf_counter()
            with ThreadPoolExecutor(ncores) as ex:
                list(ex.map(lambda i: oracle.run(hs, ys, v_begin=int(edges[i]), v_end=int(edges[i + 1])), range(ncores)))
            cpu_all_s = time.perf_counter() - c0
            got = prob.results()
            gpu = {k: (v[:, :ns] if v.ndim == 2 else v[:ns]) for k, v in got.items()}
            # Parity on the sample. The bi-exponential fit is chaotic: compare the
            # GPU-vs-CPU agreement with the agreement of two CPU builds of the same oracle source.
            nf = min(ns, 4096)
            hf, _ = make_problem(w, nf, 1, bool(args.need_f))
            yf = np.ascontiguousarray(ys[:, :nf])
            ref_f = {k: (v[:, :nf] if v.ndim == 2 else v[:nf]) for k, v in ref.items() if isinstance(v, np.ndarray)}
            floor = parity.population_stats(hf, ref_f, oracle.run_fma(hf, yf))
            stats = parity.population_stats(hs, ref, gpu)
            tolist = lambda d: {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in d.items()}
            truth = None
            if args.workload == "c3" and not args.need_f:
                # the fit is chaotic per voxel: every fp64 implementation is measured against the binary128
                # evaluation of the reference algorithm on the committed 4096-voxel sample (tests/golden/)
                sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
                import make_c3_truth as mt
                from fabber_core_amd import hiplib as _hl
                tr = parity.load_c3_truth()
                ht, yt = mt.problem(tr["n_voxels"])
                _hl.set_variant("lane")
                try:
                    g_t = _hl.run_host(ht, yt)
                finally:
                    _hl.set_variant(args.variant)
                truth = {"sample": "tests/golden/c3_truth_binary128.npz: %d voxels of this workload" % tr["n_voxels"],
                         "error_definition": "per voxel max over parameters + noise of |mean - truth| / max(|truth|, posterior sd); "
                                             "*_rel: |mean - truth| / max(|truth|, 1e-12) over the model parameters (SURVEY 8d)",
                         "gpu": parity.truth_stats(ht, tr, g_t), "cpu": parity.truth_stats(ht, tr, oracle.run(ht, yt)),
                         "cpu_fma_build": parity.truth_stats(ht, tr, oracle.run_fma(ht, yt))}
            cpu = {"value": ns / cpu_s, "unit": "voxels/s", "cores": 1, "kind": "port",
                   "sample": "first %d voxels of rank 0's shard, same model/iterations, oracle/liboracle.so single thread, %.1f s" % (ns, cpu_s),
                   "host_cpus": os.cpu_count(),
                   "all_cores": {"value": ns / cpu_all_s, "unit": "voxels/s", "cores": ncores,
                                 "gpu_over_cpu": (V / (k_ms * 1e-3)) / (ns / cpu_all_s)},
                   "gpu_over_cpu_single_thread": (V / (k_ms * 1e-3)) / (ns / cpu_s),
                   "posterior_mean_error_definition": "max over parameters+noise of |d mean| / max(|mean|, posterior sd), per voxel",
                   "gpu_vs_cpu": tolist(stats), "cpu_vs_cpu_fma_build_floor": tolist(floor),
                   "against_binary128_ground_truth": truth}
        result = {
            "metric": "voxels/sec to VB convergence", "value": value, "unit": "voxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["desc"], "voxels_per_gpu": V, "total_voxels": total_voxels, "timepoints": T,
                       "params": P, "iterations": w["its"], "need_f": bool(args.need_f), "parallelism": "voxel-shard x%d" % world,
                       "input_dtype": "f32", "mean_iterations": mean_its, "bad_voxels": int(summ[2]),
                       **({"rehearsal": "ranks share one GPU over gloo: not a measurement"} if rehearsal else {})},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if world == 1 and not args.no_e2e:
            # outside the timed region: the boundary's own rate (host pointers / the reference's C ABI)
            result["e2e"] = bench_boundary_in_child(args, V)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
