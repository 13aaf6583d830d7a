# rocprofv3 evidence for the image-in legs: per-kernel durations with ONE batch in flight and the two
# HBM-traffic counter passes, of `python3 bench.py --no-cpu --no-streaming --no-e2e` (resident uint8 images -> [binned
# Harris] -> descriptors -> matcher -> solver; the program directly behind `--`)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--streams 1 --e2e-streams 1 --no-cpu --no-streaming --no-e2e --no-i16 --steps 12 --warmup 2 --min-region-seconds 0"
rocprofv3 --kernel-trace --stats -d gpurun_out/img_stats -o s --output-format csv -- python3 bench.py $Q > gpurun_out/img_stats_bench.json 2>gpurun_out/img_stats.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/img_fetch -o p --output-format csv -- python3 bench.py $Q > /dev/null 2>gpurun_out/img_fetch.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/img_write -o p --output-format csv -- python3 bench.py $Q > /dev/null 2>gpurun_out/img_write.err &&
VISO_PMC_COMMAND="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py $Q (two separate passes, tools/image_profile.sh)" \
VISO_PMC_WORKLOAD="the resident image-in legs of bench.py: 512 pairs of synthetic 1241x376 uint8 stereo images per batch, ONE batch in flight; keypoints given (2000 per image) and binned Harris on the device (1200 per image) -> descriptors -> matcher -> solver" \
python3 tools/pmc_to_json.py gpurun_out/img_fetch gpurun_out/img_write harris_detect_kernel > gpurun_out/${TAG}_pmc_hbm_images.json &&
cp gpurun_out/img_stats/s_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_images.csv &&
cp gpurun_out/img_stats_bench.json gpurun_out/${TAG}_bench_under_rocprof_images.json
python3 tools/kstats_table.py gpurun_out/${TAG}_kernel_stats_images.csv 12
