# per-kernel durations of the image-in legs (resident uint8 images -> [Harris] -> descriptors -> matcher -> solver)
# STREAMS=1 (default): one batch in flight, durations that fit inside the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/imgstats -o s --output-format csv -- python3 bench.py --streams ${STREAMS:-1} --no-cpu --no-streaming --no-e2e --steps 20 --warmup 3 > gpurun_out/imgstats_bench.json 2>gpurun_out/imgstats.err
python3 tools/kstats_table.py gpurun_out/imgstats/s_kernel_stats.csv 24
