# kernel trace of the end-to-end loop alone (tools/host_issue_cost.py: N batches in flight, viso_batch_run only) and a per-queue timeline window
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
rocprofv3 --kernel-trace -d gpurun_out/e2etl -o s --output-format csv -- python3 tools/host_issue_cost.py 512 0 ${1:-3} > gpurun_out/e2etl.txt 2>&1
tail -2 gpurun_out/e2etl.txt
python3 tools/trace_overlap.py gpurun_out/e2etl/s_kernel_trace.csv 0.6 0.95
python3 tools/timeline.py gpurun_out/e2etl/s_kernel_trace.csv 0.7 140 > gpurun_out/e2etl_timeline.txt
