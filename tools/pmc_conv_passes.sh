#!/bin/bash
# Counter passes over the conv micro-benchmark (one rocprofv3 run per pass, each under its own limit).
#   tools/pmc_conv_passes.sh <out-dir> <LAYERS>
set -e
out=$1; export LAYERS=$2
repo=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
p() { timeout -s KILL 90 rocprofv3 --pmc $2 --output-format csv -d $repo/$out/$1 -o p -- python3 $repo/tools/bench_conv.py > $repo/$out/$1.log 2>&1; }
p a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" &&
p b "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU" &&
p c "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" &&
p d "SQ_WAIT_INST_ANY TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_HIT_sum TCC_MISS_sum"
cd $repo
for q in a b c d; do python3 tools/pmc_kernel.py $out/$q conv_ > $out/$q.txt; rm -rf $out/$q; done
