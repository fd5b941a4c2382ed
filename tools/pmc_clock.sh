#!/bin/bash
# Effective clock and MFMA-pipe occupancy per conv kernel: GRBM_GUI_ACTIVE / 8 / duration and
# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), one layer of the UNet per run.
#   tools/pmc_clock.sh <out-dir> <layer> [<layer> ...]      (ZERO=1 in the environment: all-zero operands)
set -e
out=$1; shift
repo=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  export LAYERS=$l
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE \
    --output-format csv -d $repo/$out/$l -o p -- python3 $repo/tools/bench_conv.py > $repo/$out/$l.log 2>&1
  python3 $repo/tools/pmc_table.py $(ls $repo/$out/$l/*/*counter_collection.csv $repo/$out/$l/*counter_collection.csv 2>/dev/null | head -n 1) conv > $repo/$out/$l.txt
  rm -rf $repo/$out/$l
done
