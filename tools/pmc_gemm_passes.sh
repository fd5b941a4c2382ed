#!/bin/bash
# Counter passes over one token-GEMM shape: tools/pmc_gemm_passes.sh <out-dir> SIDE CI CO
set -e
out=$1; shift
repo=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
p() { timeout -s KILL 90 rocprofv3 --pmc $2 --output-format csv -d $repo/$out/$1 -o p -- python3 $repo/tools/bench_gemm_one.py $ARGS > $repo/$out/$1.log 2>&1; }
ARGS="$*"
p a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" &&
p b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" &&
p c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES_EQ_64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES TCC_HIT_sum TCC_MISS_sum"
cd $repo
for q in a b c; do python3 tools/pmc_kernel.py $out/$q lin_gemm > $out/$q.txt; rm -rf $out/$q; done
cat $out/a.txt $out/b.txt $out/c.txt
