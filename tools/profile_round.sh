# rocprofv3 evidence for profiles/ (run on the GPU box: gpurun -- 'bash tools/profile_round.sh [tag]'):
#   1. kernel-trace stats of the default bench run (3 batches in flight: kernels of different streams overlap)
#   2. kernel-trace stats of the same workload with ONE stream: per-kernel averages that fit inside the step
#   3./4. the two HBM-traffic counter passes (separate runs, one stream)
# then: python3 tools/pmc_to_json.py gpurun_out/prof_fetch gpurun_out/prof_write <kernel> > profiles/<round>_pmc_hbm.json
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--no-cpu --no-e2e --no-streaming --no-images --no-i16"   # matcher step only (--no-e2e also leaves the drop-in leg out)
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o s --output-format csv -- python3 bench.py --no-cpu --no-streaming --no-images --no-i16 --no-drop-in > gpurun_out/prof_stats_bench.json 2>gpurun_out/prof_stats.err &&
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats1 -o s --output-format csv -- python3 bench.py --streams 1 $Q > gpurun_out/prof_stats1_bench.json 2>gpurun_out/prof_stats1.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -o p --output-format csv -- python3 bench.py --streams 1 --steps 4 --warmup 1 --min-region-seconds 0 $Q > /dev/null 2>gpurun_out/prof_fetch.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -o p --output-format csv -- python3 bench.py --streams 1 --steps 4 --warmup 1 --min-region-seconds 0 $Q > /dev/null 2>gpurun_out/prof_write.err &&
K=$(python3 -c "import json;print(json.load(open('gpurun_out/prof_stats1_bench.json'))['roofline']['kernel'])") &&
python3 tools/pmc_to_json.py gpurun_out/prof_fetch gpurun_out/prof_write "$K" > gpurun_out/${TAG}_pmc_hbm.json &&
cp gpurun_out/prof_stats/s_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_3streams.csv &&
cp gpurun_out/prof_stats1/s_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_1stream.csv &&
cp gpurun_out/prof_stats_bench.json gpurun_out/${TAG}_bench_under_rocprof_3streams.json &&
cp gpurun_out/prof_stats1_bench.json gpurun_out/${TAG}_bench_under_rocprof_1stream.json
