#!/bin/bash
# tools/cu_mask_probe.sh: per-kernel durations of the matcher step on 4/4, 3/4, 2/4, 1/4 of the CUs (one gpurun call)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
for k in 4 3 2 1; do
  rm -rf gpurun_out/cum
  rocprofv3 --kernel-trace --stats -d gpurun_out/cum -o s --output-format csv -- python3 tools/cu_mask_probe.py $k 12 ${1:-mod4} > gpurun_out/cum.txt 2>&1
  echo "== $k of 4 (${1:-mod4}): $(tail -1 gpurun_out/cum.txt)"
  python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/cum/s_kernel_stats.csv")):
    if float(r["Percentage"]) > 1.0:
        print("   %-40s avg %8.1f us" % (r["Name"].split("(")[0][:40], float(r["AverageNs"]) / 1e3), flush=True)
PY
done
