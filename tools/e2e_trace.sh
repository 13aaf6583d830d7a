# kernel trace of the end-to-end leg (3 batches in flight) -> timeline summary of its second half
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/e2etrace -o s --output-format csv -- python3 bench.py --no-cpu --no-streaming --steps 40 > gpurun_out/e2etrace_bench.json 2>gpurun_out/e2etrace.err
python3 tools/trace_overlap.py gpurun_out/e2etrace/s_kernel_trace.csv 0.75 0.98
head -20 gpurun_out/e2etrace/s_kernel_stats.csv
