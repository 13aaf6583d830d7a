#!/bin/bash
# tools/kstats_variant.sh V [V ...]: single-stream per-kernel averages of the matcher step for matcher variants (rocprofv3
# --kernel-trace --stats of a short matcher-only bench each), in ONE gpurun call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf gpurun_out/ksv
  rocprofv3 --kernel-trace --stats -d gpurun_out/ksv -o s --output-format csv -- python3 bench.py --streams 1 --steps 20 --warmup 5 --no-cpu --no-e2e --no-streaming --no-images --matcher $v $EXTRA > gpurun_out/ksv_bench_$v.json 2>/dev/null
  python3 - "$v" <<'EOP'
import csv, json, sys
v = sys.argv[1]
print("variant", v)
for r in csv.DictReader(open("gpurun_out/ksv/s_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.3:
        print("   %-60s avg %8.1f us calls %s  %s%%" % (r["Name"].split("(")[0][:60], float(r["AverageNs"]) / 1e3, r["Calls"], r["Percentage"]), flush=True)
d = json.loads(open("gpurun_out/ksv_bench_%s.json" % v).read().strip().split("\n")[-1])
print("   step %.3f ms" % d["ms_per_step"], "fps", round(d["value"]), "overflow/step", d["roofline"].get("overflow_queries_per_step"))
EOP
done
