#!/bin/bash
# tools/alone_pmc.sh KERNEL A.so B.so ...: SQ counters of one kernel of the full pipeline ALONE on the GPU (tools/run_alone.py)
# for alternative builds of the library, in one gpurun call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
k=$1; shift
for so in "$@"; do
  export VISO_HIP_SO=$so
  rm -rf gpurun_out/apmc
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace -d gpurun_out/apmc -o p --output-format csv -- python3 ${SCRIPT:-tools/run_alone.py} ${SCRIPT_ARGS:-512 6} > /dev/null 2>gpurun_out/apmc.err
  python3 - "$k" "$so" <<'PY'
import collections, csv, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/apmc/p_counter_collection.csv")):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if sys.argv[1] in k:
        print(sys.argv[2], k.split("(")[0], {c:round(sum(x)/len(x)) for c,x in v.items()}, flush=True)
PY
done
