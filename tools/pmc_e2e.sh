cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --streams 1 --no-cpu --no-streaming --steps 4 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d gpurun_out/pmce1 -o p --output-format csv -- $B > /dev/null 2>gpurun_out/pmce1.err &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT --kernel-trace -d gpurun_out/pmce2 -o p --output-format csv -- $B > /dev/null 2>gpurun_out/pmce2.err
python3 - <<'PY'
import collections, csv, json
out={}
for d in ("gpurun_out/pmce1","gpurun_out/pmce2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d+"/p_counter_collection.csv")):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        out.setdefault(k,{}).update({c:sum(x)/len(x) for c,x in v.items()})
for k,v in out.items():
    if "ransac" in k or "inlier" in k:
        print(k); print("  ",{c:round(x) for c,x in v.items()})
PY
