"""Where a kernel's scratch traffic sits: basic blocks of one function of a gfx950 assembly listing (hipcc -S
--offload-device-only) with their instruction / scratch / fp64 counts and the loops (backward branches) they are in.

    python tools/isa_blocks.py file.s <substring of the mangled kernel name> [min scratch ops to list a block]
"""
import re, sys

path, key = sys.argv[1], sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], {"label": "entry", "first": start, "ins": []}
for i in range(start + 1, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "first": i, "ins": []}
        continue
    t = l.strip()
    if t and not t.startswith((".", ";")):
        cur["ins"].append(t)
blocks.append(cur)
index = {b["label"]: n for n, b in enumerate(blocks)}
loops = []  # (head index, tail index)
for n, b in enumerate(blocks):
    for ins in b["ins"]:
        m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in index and index[m.group(1)] <= n:
            loops.append((index[m.group(1)], n))
def count(b, pat):
    return sum(1 for i in b["ins"] if re.match(pat, i))
tot = dict(ins=0, ss=0, sl=0)
print("%-14s %6s %5s %5s %6s %6s  loops (head..tail : instructions in the loop)" % ("block", "instr", "st", "ld", "fp64", "vmem"))
for n, b in enumerate(blocks):
    ss, sl = count(b, r"scratch_store"), count(b, r"scratch_load")
    tot["ins"] += len(b["ins"]); tot["ss"] += ss; tot["sl"] += sl
    if ss + sl >= least:
        inl = [(h, t) for h, t in loops if h <= n <= t]
        desc = ", ".join("%s..%s:%d" % (blocks[h]["label"], blocks[t]["label"], sum(len(x["ins"]) for x in blocks[h:t + 1])) for h, t in sorted(inl, key=lambda x: x[1] - x[0])[:3])
        print("%-14s %6d %5d %5d %6d %6d  %s" % (b["label"], len(b["ins"]), ss, sl, count(b, r"v_\w+_f64"), count(b, r"(global|buffer)_load"), desc))
print("total: %d instructions, %d scratch stores, %d scratch loads, %d loops" % (tot["ins"], tot["ss"], tot["sl"], len(loops)))
big = sorted(set(loops), key=lambda x: -sum(len(b["ins"]) for b in blocks[x[0]:x[1] + 1]))
for h, t in big:
    body = blocks[h:t + 1]
    print("loop %s..%s: %d instructions, %d scratch st, %d scratch ld, %d fp64, %d v_exp/log-ish" % (blocks[h]["label"], blocks[t]["label"], sum(len(b["ins"]) for b in body),
        sum(count(b, r"scratch_store") for b in body), sum(count(b, r"scratch_load") for b in body), sum(count(b, r"v_\w+_f64") for b in body), sum(count(b, r"v_(exp|log|ldexp|frexp)") for b in body)))
