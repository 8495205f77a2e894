#!/usr/bin/env python3
"""One JSON for the spatial workload (C5) from the per-kernel summaries tools/pmc_summary.py writes:

    python3 tools/pmc_c5_merge.py gpurun_out/r3p/r3_pmc_c5_*.json > profiles/r3_pmc_c5.json

Per kernel: launches per run, average duration, HBM bytes per launch (FETCH_SIZE / WRITE_SIZE scaled by the
bytes-per-count the calibration kernels of the same passes gave - on gfx950 FETCH_SIZE counts 2 KiB where the
documentation says 1 KiB: the x2 correction of MI355X_MICROARCH.md, measured rather than assumed), share of wave
cycles spent waiting. `per_run` adds them up over the launches of one run: the figure bench.py --workload c5 reports as
roofline.traffic."""
import json
import sys


def main():
    kernels, runs = {}, None
    for path in sys.argv[1:]:
        d = json.load(open(path))
        if not d.get("kernel_trace"):
            continue
        cal, ctr, tr = d["calibration"], d["counters"], d["kernel_trace"]
        if "setup_kernel" in tr["name"]:
            runs = tr["calls"]  # one set-up launch per run
        fetch = ctr["FETCH_SIZE"]["mean_per_launch"] * cal["read_rows<double> FETCH_SIZE"]["bytes_per_count"]
        write = ctr["WRITE_SIZE"]["mean_per_launch"] * cal["write_rows<double> WRITE_SIZE"]["bytes_per_count"]
        k = {"name": tr["name"], "calls": tr["calls"], "avg_ms": tr["avg_ns"] / 1e6, "fetch_bytes_per_launch": fetch,
             "write_bytes_per_launch": write, "hbm_gb_per_s": (fetch + write) / tr["avg_ns"],
             "calibration_bytes_per_count": {"FETCH_SIZE": cal["read_rows<double> FETCH_SIZE"]["bytes_per_count"],
                                             "WRITE_SIZE": cal["write_rows<double> WRITE_SIZE"]["bytes_per_count"]}}
        if "SQ_WAIT_ANY" in ctr and "SQ_WAVE_CYCLES" in ctr:
            k["wait_any_over_wave_cycles"] = ctr["SQ_WAIT_ANY"]["mean_per_launch"] / ctr["SQ_WAVE_CYCLES"]["mean_per_launch"]
            k["wait_inst_over_wave_cycles"] = ctr["SQ_WAIT_INST_ANY"]["mean_per_launch"] / ctr["SQ_WAVE_CYCLES"]["mean_per_launch"]
        if "SQ_INSTS_VALU" in ctr:
            k["valu_wave_instructions"] = ctr["SQ_INSTS_VALU"]["mean_per_launch"]
        kernels[d["kernel_match"]] = k
    if not runs:
        raise SystemExit("no set-up kernel among the summaries: cannot tell how many runs the profile holds")
    total = {"fetch_bytes": 0.0, "write_bytes": 0.0, "kernel_ms": 0.0}
    for k in kernels.values():
        k["launches_per_run"] = k["calls"] / runs
        total["fetch_bytes"] += k["fetch_bytes_per_launch"] * k["launches_per_run"]
        total["write_bytes"] += k["write_bytes_per_launch"] * k["launches_per_run"]
        total["kernel_ms"] += k["avg_ms"] * k["launches_per_run"]
    total["traffic"] = total["fetch_bytes"] + total["write_bytes"]
    total["note"] = ("prep, ordered sweep, second sweep (both instances), set-up and a_K partial-sum kernels; the a_K reductions, the numbering "
                     "kernels and the copies are not counted (3 % of the run's device time)")
    json.dump({"workload": "c5", "runs_profiled": runs, "kernels": kernels, "per_run": total}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
