#!/bin/bash
# tools/experiments/dropin_swap_stats.sh KERNEL A.so B.so ...: like dropin_swap.sh, but under rocprofv3 --kernel-trace --stats: the average
# duration of one kernel of the per-call loop for each build (the GPU box's disposable copy of the tree only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
k=$1; shift
cp libviso_amd/libviso_hip.so /tmp/libviso_hip.keep
for so in "$@"; do
  cp $so libviso_amd/libviso_hip.so
  rm -rf gpurun_out/dss
  rocprofv3 --kernel-trace --stats -d gpurun_out/dss -o s --output-format csv -- python3 tools/dropin_probe.py 257 2000 > gpurun_out/dss.txt 2>&1
  python3 - "$k" "$so" <<'PY'
import csv, sys
for r in csv.DictReader(open("gpurun_out/dss/s_kernel_stats.csv")):
    if any(t in r["Name"] for t in sys.argv[1].split(",")):
        print("%s: %-28s calls %s avg %.1f us (min %.1f max %.1f)" % (sys.argv[2], r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3), flush=True)
PY
done
cp /tmp/libviso_hip.keep libviso_amd/libviso_hip.so
