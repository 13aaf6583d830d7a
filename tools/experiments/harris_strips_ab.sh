#!/bin/bash
# Round 6: the Harris detector's strip kernel against the wave-per-bin kernel, each ALONE on the GPU (tools/run_alone_images.py under
# rocprofv3 --kernel-trace --stats), alternating, one box:  gpurun -- 'bash tools/experiments/harris_strips_ab.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
for v in 0 1 0 1; do
  export VISO_HARRIS_STRIPS=$v
  rm -rf gpurun_out/hs
  rocprofv3 --kernel-trace --stats -d gpurun_out/hs -o s --output-format csv -- python3 tools/run_alone_images.py ${1:-128} 12 > gpurun_out/hs.txt 2>&1
  python3 - $v <<'PY'
import csv, sys
for r in csv.DictReader(open("gpurun_out/hs/s_kernel_stats.csv")):
    if "harris" in r["Name"]:
        print("STRIPS=%s  %-28s calls %s avg %.1f us (min %.1f)" % (sys.argv[1], r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3), flush=True)
PY
done
