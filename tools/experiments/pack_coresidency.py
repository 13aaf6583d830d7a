"""How much of pack_desc_kernel runs BESIDE another batch's match_union8_kernel when several batches are in flight (a
rocprofv3 --kernel-trace CSV of bench.py's matcher-only leg): per pack launch the share of its duration during which a
match_union8_kernel of another queue was running, and both kernels' durations against their durations alone.
    python3 tools/experiments/pack_coresidency.py s_kernel_trace.csv [from_fraction to_fraction]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
rows.sort()
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 0.98)
t0, t1 = rows[0][0], max(r[1] for r in rows)
a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
sel = [r for r in rows if r[0] >= a and r[1] <= b]
u8 = [r for r in sel if "match_union8" in r[2]]
pk = [r for r in sel if r[2].startswith("pack_desc")]
st = [r for r in sel if "match_stereo" in r[2]]


def overlap(x, others):
    tot = 0
    for o in others:
        if o[3] == x[3]:
            continue
        tot += max(0, min(x[1], o[1]) - max(x[0], o[0]))
    return tot


def summary(name, ks, others, oname):
    if not ks:
        return
    d = sum(k[1] - k[0] for k in ks)
    ov = sum(min(overlap(k, others), k[1] - k[0]) for k in ks)
    print("%-22s %4d launches, %.3f ms on average, %.0f %% of that beside a %s of another queue" % (name, len(ks), d / len(ks) / 1e6, 100.0 * ov / d, oname))


summary("pack_desc_kernel", pk, u8, "match_union8_kernel")
summary("match_union8_kernel", u8, pk, "pack_desc_kernel")
summary("match_union8_kernel", u8, u8, "match_union8_kernel")
summary("match_stereo_kernel", st, u8, "match_union8_kernel")
wall = max(r[1] for r in sel) - sel[0][0]
steps = len(u8)
print("window %.3f ms, %d steps: %.3f ms per step; queues %d" % (wall / 1e6, steps, wall / 1e6 / max(steps, 1), len(set(r[3] for r in sel))))
