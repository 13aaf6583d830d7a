# every kernel of the full pipeline ALONE (one batch, synchronize per run): durations, then SQ counters of the RANSAC stage
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
rocprofv3 --kernel-trace --stats -d gpurun_out/alone -o s --output-format csv -- python3 tools/run_alone.py ${1:-512} 12 > gpurun_out/alone.txt 2>&1
tail -1 gpurun_out/alone.txt
python3 tools/kstats_table.py gpurun_out/alone/s_kernel_stats.csv 24
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace -d gpurun_out/alonepmc -o p --output-format csv -- python3 tools/run_alone.py ${1:-512} 6 > /dev/null 2>gpurun_out/alonepmc.err
python3 - <<'PY'
import collections, csv
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/alonepmc/p_counter_collection.csv")):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "ransac" in k or "inlier" in k or "circle" in k:
        print(k.split("(")[0], {c:round(sum(x)/len(x)) for c,x in v.items()})
PY
