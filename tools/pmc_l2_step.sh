#!/bin/bash
# L2 hit / miss requests per kernel over a few steps of a bench workload: tools/pmc_l2_step.sh <out-dir> <workload>
set -e
out=$1; w=$2
repo=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $repo/$out/l2 -o p -- python3 $repo/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-small-batch --no-e2e --no-roofline > $repo/$out/l2.log 2>&1
cd $repo
python3 tools/pmc_kernel.py $out/l2 > $out/l2_$w.txt
rm -rf $out/l2
