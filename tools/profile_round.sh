# rocprofv3 evidence for profiles/: kernel-trace stats of the default bench run, then the two HBM-traffic
# counter passes (separate runs, one stream so that dispatches are not concurrent).  Run on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh'
# then: python3 tools/pmc_to_json.py gpurun_out/prof_fetch gpurun_out/prof_write > profiles/<round>_pmc.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o s --output-format csv -- python3 bench.py > gpurun_out/prof_stats_bench.json 2>gpurun_out/prof_stats.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -o p --output-format csv -- python3 bench.py --streams 1 --steps 4 --warmup 1 --no-cpu --no-e2e > /dev/null 2>gpurun_out/prof_fetch.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -o p --output-format csv -- python3 bench.py --streams 1 --steps 4 --warmup 1 --no-cpu --no-e2e > /dev/null 2>gpurun_out/prof_write.err
ls gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
