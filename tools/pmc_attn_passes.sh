#!/bin/bash
# Counter passes over the window-attention micro-benchmark: tools/pmc_attn_passes.sh <out-dir>
set -e
out=$1
repo=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
p() { timeout -s KILL 90 rocprofv3 --pmc $2 --output-format csv -d $repo/$out/$1 -o p -- python3 $repo/tools/bench_attn.py > $repo/$out/$1.log 2>&1; }
p a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" &&
p b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" &&
p c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT TCC_HIT_sum TCC_MISS_sum"
cd $repo
for q in a b c; do python3 tools/pmc_kernel.py $out/$q attn_bwd > $out/$q.txt; rm -rf $out/$q; done
cat $out/a.txt $out/b.txt $out/c.txt
