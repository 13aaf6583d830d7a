#!/bin/bash
# tools/experiments/env_stats.sh KERNELS "ENV=a" "ENV=b" ...: average durations of some kernels of the per-call loop (rocprofv3 --kernel-trace
# --stats around tools/dropin_probe.py) for each setting of an environment knob of the product library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
k=$1; shift
for e in "$@"; do
  rm -rf gpurun_out/ess
  env $e rocprofv3 --kernel-trace --stats -d gpurun_out/ess -o s --output-format csv -- python3 tools/dropin_probe.py 257 2000 > gpurun_out/ess.txt 2>&1
  python3 - "$k" "$e" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open("gpurun_out/ess/s_kernel_stats.csv")):
    if any(t in r["Name"] for t in sys.argv[1].split(",")):
        tot += float(r["AverageNs"]) / 1e3
        print("%s: %-28s calls %s avg %.1f us (min %.1f max %.1f)" % (sys.argv[2], r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3), flush=True)
print("%s: sum of the averages %.1f us" % (sys.argv[2], tot), flush=True)
PY
done
