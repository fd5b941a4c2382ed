#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   kernel-trace stats of the default bench command, and separate --pmc passes for HBM traffic
#   (FETCH_SIZE, WRITE_SIZE) and MFMA busy cycles, for the train / predict / swin workloads.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# the default command, workload by workload (the same timed regions; one stats file per workload)
for w in train swin predict; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_trace -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 --settle-seconds 0 --no-cpu-baseline --no-small-batch --no-e2e > $O/${w}_bench.log 2>&1
done
for w in train predict swin; do
  extra="--steps 2 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-roofline --no-e2e"
  [ $w = predict ] && extra="--steps 1 --warmup 0 --settle-seconds 0 --no-cpu-baseline --no-roofline --no-e2e --grid 4096"
  [ $w = swin ] && extra="--steps 2 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-roofline --no-small-batch --no-e2e"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${w}_fetch -- python3 $R/bench.py --workload $w $extra > $O/${w}_fetch.log 2>&1
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${w}_write -- python3 $R/bench.py --workload $w $extra > $O/${w}_write.log 2>&1
  timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${w}_mfma -- python3 $R/bench.py --workload $w $extra > $O/${w}_mfma.log 2>&1
done
echo collected
