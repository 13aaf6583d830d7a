#!/bin/bash
# the image-in pipeline's kernels ALONE on the GPU (one 128-pair batch, synchronize per run: tools/run_alone_images.py):
# durations (rocprofv3 --kernel-trace --stats) and SQ counters (rocprofv3 --pmc) of the detector, the extractor and the
# kernels behind them -> profiles/<tag>_image_kernels_alone.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
echo "# the image-in pipeline (binned Harris on the device -> descriptors -> matcher -> solver), ONE 128-pair batch, synchronize per run: tools/run_alone_images.py"
echo "# durations: rocprofv3 --kernel-trace --stats, proper CSV (tools/kstats_table.py); counters: rocprofv3 --pmc SQ_*, per launch"
rocprofv3 --kernel-trace --stats -d gpurun_out/ialone -o s --output-format csv -- python3 tools/run_alone_images.py 128 12 > gpurun_out/ialone.txt 2>&1
python3 tools/kstats_table.py gpurun_out/ialone/s_kernel_stats.csv 24
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace -d gpurun_out/ialonepmc -o p --output-format csv -- python3 tools/run_alone_images.py 128 6 > /dev/null 2>gpurun_out/ialonepmc.err
python3 - <<'PY'
import collections, csv
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/ialonepmc/p_counter_collection.csv")):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if any(t in k for t in ("harris", "extract", "match_union8", "match_stereo", "ransac", "inlier", "circle")):
        print(k.split("(")[0], {c:round(sum(x)/len(x)) for c,x in v.items()})
PY
