# two quick counter passes over the matcher-only default workload, one stream (development aid):
#   gpurun -- 'VISO_BENCH_EXTRA="--matcher 5" bash tools/pmc_quick.sh'  -> prints the temporal kernel's counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --streams 1 --no-cpu --no-e2e --no-streaming --no-images --steps 3 --warmup 1 --min-region-seconds 0 $VISO_BENCH_EXTRA"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmcq1 -o p --output-format csv -- $B > /dev/null 2>gpurun_out/pmcq1.err &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d gpurun_out/pmcq2 -o p --output-format csv -- $B > /dev/null 2>gpurun_out/pmcq2.err &&
python3 - <<PY
import collections, csv
for d in ("gpurun_out/pmcq1", "gpurun_out/pmcq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "match_prune" in k or "match_union" in k:
            print(k.split("(")[0], {c: round(sum(x) / len(x) / 1e6, 2) for c, x in sorted(v.items())}, "(millions per launch)")
PY
