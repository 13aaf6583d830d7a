#!/bin/bash
# Round 6, VERDICT r5 item 2 sized before anything is rebuilt: pack_desc_kernel WITHOUT its u16-row store (a -DVISO_DEBUG_VARIANTS
# build, $VISO_EXP_PACK_NO_U16=1: the results are garbage, only the pack kernel's time means anything) against the same build
# with it: what "planes only" can buy the pack kernel at the very most.  gpurun -- 'bash tools/experiments/pack_no_u16.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=. VISO_HIP_SO=$GRAFT_REPO_ROOT/build_ab/dbg.so
for v in 0 1 0 1; do
  export VISO_EXP_PACK_NO_U16=$v
  rm -rf gpurun_out/pk
  rocprofv3 --kernel-trace --stats -d gpurun_out/pk -o s --output-format csv -- python3 tools/host_issue_cost.py 512 0 1 0 0 matcher > gpurun_out/pk.txt 2>&1
  python3 - $v <<'PY'
import csv, sys
for r in csv.DictReader(open("gpurun_out/pk/s_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("pack_desc_kernel", "match_union8", "match_stereo")):
        print("NO_U16=%s  %-28s calls %s avg %.1f us" % (sys.argv[1], r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3), flush=True)
PY
done
