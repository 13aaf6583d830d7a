#!/bin/bash
# tools/alone_ab.sh KERNEL A.so B.so ...: duration of one kernel of the full pipeline ALONE on the GPU (tools/run_alone.py under
# rocprofv3 --kernel-trace --stats) for alternative builds of the library, in one gpurun call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
k=$1; shift
for so in "$@"; do
  export VISO_HIP_SO=$so
  rm -rf gpurun_out/aab
  rocprofv3 --kernel-trace --stats -d gpurun_out/aab -o s --output-format csv -- python3 ${SCRIPT:-tools/run_alone.py} ${SCRIPT_ARGS:-512 12} > gpurun_out/aab.txt 2>&1
  python3 - "$k" "$so" <<'PY'
import csv, sys
for r in csv.DictReader(open("gpurun_out/aab/s_kernel_stats.csv")):
    if sys.argv[1] in r["Name"]:
        print("%s: %-32s calls %s avg %.1f us (min %.1f max %.1f)" % (sys.argv[2], r["Name"].split("(")[0][:32], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3), flush=True)
PY
done
