# per-kernel averages of one 512-pair batch alone, matcher only: bash tools/kstats.sh [frames]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
rocprofv3 --kernel-trace --stats -d gpurun_out/kstats -o s --output-format csv -- python3 tools/host_issue_cost.py ${1:-512} 0 1 0 0 matcher > gpurun_out/kstats.txt 2>&1
head -8 gpurun_out/kstats/s_kernel_stats.csv | cut -c1-100
python3 tools/host_issue_cost.py ${1:-512} 0 3 0 0 matcher
