#!/bin/bash
# tools/profile_all.sh [tag]: everything profiles/<tag>_* is made of, in ONE gpurun call (about ten minutes of GPU time):
# kernel stats (3 streams / 1 stream), HBM and SQ counter passes, the full pipeline's kernel stats, the image legs, and the
# bench lines (default with CPU baseline, RCCL group of one, clustered, configs[4] geometry).  Results: gpurun_out/<tag>_*
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1 && echo "profile_round ok" &&
bash tools/pmc_batch.sh $TAG > gpurun_out/${TAG}_pmc_batch.log 2>&1 && echo "pmc_batch ok" &&
bash tools/solver_stats.sh 512 > gpurun_out/${TAG}_solver_stats.log 2>&1 && cp gpurun_out/solverstats/s_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_e2e_1stream.csv && echo "solver_stats ok" &&
bash tools/image_profile.sh $TAG > gpurun_out/${TAG}_image_profile.log 2>&1 && echo "image_profile ok" &&
bash tools/ransac_alone.sh 512 > gpurun_out/${TAG}_ransac_alone.txt 2>gpurun_out/${TAG}_ransac_alone.err && echo "ransac_alone ok" &&
bash tools/image_alone.sh > gpurun_out/${TAG}_image_kernels_alone.txt 2>gpurun_out/${TAG}_image_alone.err && echo "image_alone ok" &&
python3 tools/dropin_probe.py 257 2000 > gpurun_out/${TAG}_drop_in.txt 2>&1 && echo "drop_in ok" &&
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/dropin_stats -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dropin_probe.py 257 2000 > $GRAFT_REPO_ROOT/gpurun_out/dropin_stats.txt 2>&1) && cp gpurun_out/dropin_stats/s_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_drop_in.csv && echo "drop_in stats ok" &&
./tools/h2d_probe > gpurun_out/${TAG}_h2d_probe.txt 2>&1 && echo "h2d_probe ok" &&
python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err && echo "bench default ok" &&
python3 bench.py --force-collective --no-cpu --no-streaming --no-images > gpurun_out/${TAG}_bench_rccl_1rank.json 2> gpurun_out/${TAG}_bench_rccl.err && echo "bench rccl ok" &&
python3 bench.py --clustered 0.7 --no-cpu --no-streaming --no-images > gpurun_out/${TAG}_bench_clustered.json 2> /dev/null && echo "bench clustered ok" &&
python3 bench.py --kp 8000 --width 2048 --height 1024 --frames 128 --steps 20 --warmup 3 --no-cpu --no-streaming --no-images > gpurun_out/${TAG}_bench_config4.json 2> /dev/null && echo "bench config4 ok" &&
VISO_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --backend gloo --no-cpu --no-streaming --no-images > gpurun_out/${TAG}_bench_2rank_gloo.json 2> /dev/null && echo "bench 2rank ok"
