# SQ / TA / TCP counter passes over the default matcher workload, one stream (bench.py, matcher only):
#   gpurun -- 'bash tools/pmc_batch.sh [tag]'   ->  gpurun_out/<tag>_pmc_sq.json (per-kernel averages)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --streams 1 --no-cpu --no-e2e --no-streaming --no-images --no-i16 --steps 3 --warmup 1 --min-region-seconds 0 $VISO_BENCH_EXTRA"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d gpurun_out/pmcb1 -o p --output-format csv -- $B > /dev/null 2>gpurun_out/pmcb1.err &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/pmcb2 -o p --output-format csv -- $B > /dev/null 2>gpurun_out/pmcb2.err &&
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmcb3 -o p --output-format csv -- $B > /dev/null 2>gpurun_out/pmcb3.err &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmcb4 -o p --output-format csv -- $B > /dev/null 2>gpurun_out/pmcb4.err &&
python3 tools/pmc_summary.py gpurun_out/pmcb1 gpurun_out/pmcb2 gpurun_out/pmcb3 gpurun_out/pmcb4 > gpurun_out/${TAG}_pmc_sq.json   # the last pass wins: VALUBusy's three counters come from ONE run
