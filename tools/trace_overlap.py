"""Timeline summary of a rocprofv3 --kernel-trace CSV: wall time, time with at least one kernel running,
sum of kernel durations, and the per-kernel share of the busy time — to see how well the streams overlap."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
lo_frac, hi_frac = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
t0, t1 = rows[0][0], max(r[1] for r in rows)
a, b = t0 + (t1 - t0) * lo_frac, t0 + (t1 - t0) * hi_frac
sel = [r for r in rows if r[0] >= a and r[1] <= b]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = max(r[1] for r in sel) - sel[0][0]
tot = collections.Counter()
for s, e, k, _ in sel:
    tot[k.split("(")[0][:40]] += e - s
print(f"window {wall/1e6:.3f} ms, >=1 kernel running {busy/1e6:.3f} ms ({100*busy/wall:.1f} %), sum of durations {sum(tot.values())/1e6:.3f} ms, queues {len(set(r[3] for r in sel))}")
for k, v in tot.most_common(10):
    print(f"  {k:42s} {v/1e6:8.3f} ms")
