# per-kernel averages of one batch alone (full pipeline), then the cost of each solver kernel with three batches in flight
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
rocprofv3 --kernel-trace --stats -d gpurun_out/solverstats -o s --output-format csv -- python3 tools/host_issue_cost.py 256 0 1 > gpurun_out/solverstats.txt 2>&1
head -20 gpurun_out/solverstats/s_kernel_stats.csv | cut -c1-100
for k in x c hc i r; do echo skip $k; VISO_EXP_SKIP=$k python tools/host_issue_cost.py 256 0 3; done
for hb in 64 256; do echo hyp block $hb; VISO_EXP_HYP_BLOCK=$hb python tools/host_issue_cost.py 256 0 3; done
