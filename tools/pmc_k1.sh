#!/bin/bash
# PMC passes of the C3 preprocessing tool (run on the GPU box from the repo root):  bash tools/pmc_k1.sh gpurun_out/pmc_k1
# pass 1: SQ issue / LDS counters; pass 2: FETCH_SIZE; pass 3: WRITE_SIZE (separate passes, kernel trace only in none of them)
set -e
OUT=$(realpath -m "$1"); R=$(pwd); export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
T="python3 $R/tools/bench_c3.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq" -- $T > "$OUT/sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $T > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $T > "$OUT/write.log" 2>&1
cd "$R"
python3 tools/pmc_summary.py "$OUT/sq" | grep -i "kernel\|resize\|resample" > "$OUT/sq_summary.csv"
python3 tools/pmc_summary.py "$OUT/fetch" "$OUT/write" | grep -i "kernel\|resize\|resample" > "$OUT/traffic_summary.csv"
cat "$OUT/sq_summary.csv" "$OUT/traffic_summary.csv"
