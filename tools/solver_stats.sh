# per-kernel averages of one batch alone (full pipeline)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
rocprofv3 --kernel-trace --stats -d gpurun_out/solverstats -o s --output-format csv -- python3 tools/host_issue_cost.py ${1:-512} 0 1 > gpurun_out/solverstats.txt 2>&1
python3 tools/kstats_table.py gpurun_out/solverstats/s_kernel_stats.csv 22
