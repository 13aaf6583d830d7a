"""How well the heavy (throughput-bound) kernels of the step pack: share of the wall time with at least one of them
running, their mean stretch, and what runs in the gaps.  Usage: python tools/heavy_busy.py trace.csv [lo_frac hi_frac]"""
import collections
import csv
import sys

HEAVY = ("match_union",  # match_union_kernel and match_union8_kernel (the default since round 4)
         "match_prune_kernel", "pack_desc", "match_stereo_kernel", "extract_pack", "harris")
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:34], r["Queue_Id"]))
rows.sort()
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 0.95)
t0, t1 = rows[0][0], max(r[1] for r in rows)
a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
sel = [r for r in rows if r[0] >= a and r[1] <= b]


def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)


wall = max(r[1] for r in sel) - sel[0][0]
heavy = [(s, e) for s, e, k, _ in sel if k.startswith(HEAVY)]
n_steps = sum(1 for _, _, k, _ in sel if k.startswith(("match_union", "match_prune_kernel")))
print(f"window {wall / 1e6:.3f} ms, {n_steps} steps -> {wall / 1e3 / max(n_steps, 1):.1f} us per step")
print(f"  >= 1 heavy kernel running: {100 * union(heavy) / wall:.1f} % of the wall time; any kernel: {100 * union([(s, e) for s, e, _, _ in sel]) / wall:.1f} %")
dur = collections.defaultdict(list)
for s, e, k, _ in sel:
    dur[k].append(e - s)
print("  mean duration (us), per step total:")
tot = 0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"    {k:36s} {sum(v) / len(v) / 1e3:8.1f}  x{len(v)}")
    tot += sum(v) / max(n_steps, 1)
print(f"  sum of mean durations per step {tot / 1e3:.1f} us")
