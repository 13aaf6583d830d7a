"""Selected columns of a rocprofv3 `*_kernel_stats.csv` as a proper CSV (csv module on both sides: kernel names contain
commas, `cut -d,` splits inside them -- that is how profiles/r04_ransac_alone.txt was mangled in round 4).
Usage: python tools/kstats_table.py stats.csv [max_rows]"""
import csv
import sys

COLS = ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs")
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC, lineterminator="\n")
w.writerow(COLS)
for r in rows[:n]:
    w.writerow([r["Name"], int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), int(r["MinNs"]), int(r["MaxNs"])])
