#!/usr/bin/env python3
"""Summarise the passes of tools/pmc_passes.sh into one JSON (per-launch averages for the VB
kernel, calibration factors for FETCH_SIZE / WRITE_SIZE at the kernels' access widths).

    python tools/pmc_summary.py gpurun_out/pmc_<tag> [kernel-substring] > profiles/<name>.json
"""
import csv
import glob
import json
import os
import sys


def counters(d, match, durations=None):
    """{counter: [value per dispatch]} for kernels whose name contains `match`; `durations` (a list) receives the
    duration in ns of every such dispatch IN THIS PASS (the counter rows carry the dispatch's own timestamps)."""
    out = {}
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        per, span = {}, {}
        for row in csv.DictReader(open(f)):
            if match not in row["Kernel_Name"]:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
            if row.get("End_Timestamp"):
                span[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        for (disp, name), val in per.items():
            out.setdefault(name, []).append(val)
        if durations is not None:
            durations.extend(span.values())
    return out


def mean(xs):
    return sum(xs) / len(xs) if xs else None


def kernel_stats(d, match):
    for f in glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            if match in row["Name"]:
                return dict(name=row["Name"], calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]),
                            min_ns=float(row["MinNs"]), max_ns=float(row["MaxNs"]), percentage=float(row["Percentage"]))
    return None


def main():
    root = sys.argv[1]
    match = sys.argv[2] if len(sys.argv) > 2 else "vb_lane_kernel"
    out = {"kernel_match": match}
    out["kernel_trace"] = kernel_stats(os.path.join(root, "trace"), match)
    # calibration: bytes actually moved / counter value, per access pattern (counter unit: KB? -> derive)
    expect = {"read_rows<float>": 512 * (1 << 20) * 4, "read_rows<double>": 512 * (1 << 20) * 8,
              "write_rows<float>": 512 * (1 << 20) * 4, "write_rows<double>": 512 * (1 << 20) * 8}
    calib = {}
    for which, ctr in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
        for kern, nbytes in expect.items():
            vals = counters(os.path.join(root, which), kern.replace("<", "I").split("I")[0]) if False else None
        f = glob.glob(os.path.join(root, which, "*", "*_counter_collection.csv"))
        if not f:
            continue
        per = {}
        for row in csv.DictReader(open(f[0])):
            if row["Counter_Name"] != ctr:
                continue
            key = (row["Kernel_Name"], row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
        byk = {}
        for (kn, _), v in per.items():
            byk.setdefault(kn, []).append(v)
        for kn, vals in byk.items():
            for pat, nbytes in expect.items():
                base, typ = pat.split("<")
                typ = typ.rstrip(">")
                if base in kn and (("IfE" in kn or "<float>" in kn) if typ == "float" else ("IdE" in kn or "<double>" in kn)):
                    if (ctr == "FETCH_SIZE") == base.startswith("read"):
                        calib[pat + " " + ctr] = dict(counter=mean(vals), true_bytes=nbytes, bytes_per_count=nbytes / mean(vals))
    out["calibration"] = calib
    res, pass_ns = {}, {}
    for which in ("fetch", "write", "sq", "fp64", "mem"):
        dur = []
        for name, vals in counters(os.path.join(root, which), match, dur).items():
            res[name if name not in res else name + "@" + which] = dict(mean_per_launch=mean(vals), launches=len(vals), counter_pass=which)
        if dur:
            pass_ns[which] = dict(avg_ns=mean(dur), launches=len(dur))
    out["counters"] = res
    out["pass_durations"] = pass_ns  # the kernel's duration inside each counter pass (counters slow a kernel down a little)
    # derived: shader clock held during the kernel, executed fp64 FLOP/s, VALU issue utilisation at that clock
    kt = out["kernel_trace"]
    g = lambda n: res.get(n, {}).get("mean_per_launch")
    gui = [k for k in res if k.startswith("GRBM_GUI_ACTIVE") and res[k]["counter_pass"] == "fp64"]
    if kt and gui and "fp64" in pass_ns:
        # Every derived figure divides a pass's counters by the kernel's duration IN THAT SAME PASS (round-3 verdict:
        # a fraction built from one run's counters and another run's clock is not a measurement of either).
        t_fp64 = pass_ns["fp64"]["avg_ns"]
        clock_ghz = res[gui[0]]["mean_per_launch"] / 8 / t_fp64
        d = {"shader_clock_ghz": clock_ghz, "fp64_pass_avg_ns": t_fp64, "trace_pass_avg_ns": kt["avg_ns"]}
        if g("SQ_INSTS_VALU_FMA_F64") is not None:
            flop = (2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64")) * 64
            d["fp64_flop_per_launch"] = flop
            d["fp64_tflops"] = flop / t_fp64 / 1e3
            d["fp64_vector_peak_tflops_at_2p4ghz"] = 78.6
            d["fp64_frac_of_peak"] = d["fp64_tflops"] / 78.6
        gui_sq = [k for k in res if k.startswith("GRBM_GUI_ACTIVE") and res[k]["counter_pass"] == "sq"]
        if g("SQ_INSTS_VALU") and gui_sq:
            # cycles of the sq pass itself: GRBM_GUI_ACTIVE is summed over the 8 XCDs
            d["valu_issue_utilisation"] = g("SQ_INSTS_VALU") * 4 / (1024 * res[gui_sq[0]]["mean_per_launch"] / 8)
        elif g("SQ_INSTS_VALU") and "sq" in pass_ns:
            d["valu_issue_utilisation"] = g("SQ_INSTS_VALU") * 4 / (1024 * pass_ns["sq"]["avg_ns"] * clock_ghz)
        out["derived"] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
