#!/bin/bash
# Condenses what tools/collect_profiles.sh left under gpurun_out/<tag>/ into the summaries kept in profiles/.
#   bash tools/freeze_profiles.sh <tag under gpurun_out> <prefix under profiles, e.g. r02a> [output directory]
# (the raw traces of one collection exceed what a GPU call copies back: run it on the GPU box right after the
#  collection with an output directory under gpurun_out/, delete the raw data there, and move the summaries into
#  profiles/ afterwards)
set -e
cd "$(dirname "$0")/.."
G=gpurun_out/$1; D=${3:-profiles}; P=$D/$2
mkdir -p $D
declare -A TITLE=(
 [train]="rocprofv3 --kernel-trace --stats -- python3 bench.py --workload train --steps 10 --warmup 3 (13 steps; UNet(4,1,48) b64 4x256x256 bf16: fwd + loss(all) + bwd + clip + AdamW)"
 [swin]="rocprofv3 --kernel-trace --stats -- python3 bench.py --workload swin --steps 10 --warmup 3 (13 eager steps; SwinUNet b64 4x256x256 bf16 trainer step: fwd + bilinear x2 + loss(tid_gradient) + bwd + clip + AdamW)"
 [predict]="rocprofv3 --kernel-trace --stats -- python3 bench.py --workload predict (1 warm-up + 3 timed passes over a 8192x8192 grid, 3969 tiles each, UNet(4,1,48) eval bf16)")
declare -A STEPS=([train]=13 [swin]=13 [predict]=4)
for w in train swin predict; do
  python tools/prof_summary.py $G/${w}_trace/vm/*_kernel_stats.csv ${P}_${w}_kernel_stats.csv "${TITLE[$w]}" ${STEPS[$w]}
  python tools/roofline_table.py $G/${w}_fetch/vm/*_counter_collection.csv $G/${w}_write/vm/*_counter_collection.csv \
      $G/${w}_mfma/vm/*_counter_collection.csv ${P}_${w}_pmc_roofline.csv \
      "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE), bench.py --workload $w"
done
python tools/pmc_traffic.py $G/train_fetch/vm/*_counter_collection.csv $G/train_write/vm/*_counter_collection.csv $D/pmc_traffic.json $G/train_mfma/vm/*_counter_collection.csv
python tools/pmc_traffic.py $G/swin_fetch/vm/*_counter_collection.csv $G/swin_write/vm/*_counter_collection.csv $D/pmc_traffic_swin.json $G/swin_mfma/vm/*_counter_collection.csv
