#!/bin/bash
# tools/kernel_ab.sh KERNEL_SUBSTRING A.so B.so ...: single-stream average duration of one kernel for several builds of the
# library (rocprofv3 --kernel-trace --stats of a short matcher-only bench each), in ONE gpurun call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
k=$1; shift
for so in "$@"; do
  VISO_HIP_SO=$so rocprofv3 --kernel-trace --stats -d gpurun_out/kab -o s --output-format csv -- python3 bench.py --streams 1 --steps 20 --warmup 5 --no-cpu --no-e2e --no-streaming --no-images $EXTRA > gpurun_out/kab_bench.json 2>/dev/null
  python3 - "$so" "$k" <<'EOP'
import csv, json, sys
so, k = sys.argv[1], sys.argv[2]
for r in csv.DictReader(open("gpurun_out/kab/s_kernel_stats.csv")):
    if k in r["Name"]:
        print(so, r["Name"].split("(")[0], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "calls", r["Calls"], flush=True)
d = json.loads(open("gpurun_out/kab_bench.json").read().strip().split("\n")[-1])
print("   step %.3f ms" % d["ms_per_step"], "fps", round(d["value"]))
EOP
done
