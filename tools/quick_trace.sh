#!/bin/bash
# One workload under rocprofv3 --kernel-trace --stats, condensed table on stdout (GPU box, repo root):
#   bash tools/quick_trace.sh train|swin|predict <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
w=${1:-train}; O=$R/gpurun_out/${2:-qt}_$w
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-small-batch > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
steps=13; [ $w = predict ] && steps=4
cd $R && python tools/prof_summary.py $O/trace/*/*_kernel_stats.csv $O/summary.csv "$w" $steps | cut -c1-110
rm -rf $O/trace
