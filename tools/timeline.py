"""Print a window of a rocprofv3 --kernel-trace CSV as a timeline: start, end, duration (us), queue, kernel.
Usage: python tools/timeline.py trace.csv [start_frac=0.9] [n_rows=100]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:30], r["Queue_Id"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
t0, t1 = rows[0][0], rows[-1][1]
sel = [r for r in rows if r[0] >= t0 + (t1 - t0) * frac][:n]
b = sel[0][0]
for s, e, k, q in sel:
    print(f"{(s - b) / 1e3:9.1f} {(e - b) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q:3s} {k}")
