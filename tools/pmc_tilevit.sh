#!/bin/bash
# PMC passes of the tile-ViT bench tool (run on the GPU box from the repo root):  bash tools/pmc_tilevit.sh gpurun_out/pmc_tilevit [images]
# pass kt: kernel trace + stats; pass sq1 / sq2: SQ issue / wait / LDS counters (separate passes, no trace domains beside --pmc)
set -e
OUT=$(realpath -m "$1"); N=${2:-2}; R=$(pwd); export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
T="python3 $R/tools/bench_tilevit.py $N"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $T > "$OUT/kt.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/sq1" -- $T > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_LDS --output-format csv -d "$OUT/sq2" -- $T > "$OUT/sq2.log" 2>&1
cd "$R"
python3 tools/pmc_summary.py "$OUT/sq1" | grep -i "kernel\|attn" > "$OUT/sq1_summary.csv"
python3 tools/pmc_summary.py "$OUT/sq2" | grep -i "kernel\|attn" > "$OUT/sq2_summary.csv"
cat "$OUT/sq1_summary.csv" "$OUT/sq2_summary.csv"
find "$OUT/kt" -name "*kernel_stats.csv" | head -1 | xargs head -6
