#!/bin/bash
# The five rocprofv3 passes of the headline bench command (run on the GPU box, from the repo root):
#   bash tools/collect_profiles.sh gpurun_out/prof_r2
# then:  python3 tools/profile_record.py gpurun_out/prof_r2 round2_vN
set -e
OUT=$(realpath -m "$1")
R=$(pwd)
export TMPDIR=/tmp
rm -rf "$OUT"  # gpurun merges into an existing gpurun_out/: stale pass files of an earlier call would be folded in twice
mkdir -p "$OUT"
cd /tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/sq" -- $BENCH > "$OUT/sq.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/grbm" -- $BENCH > "$OUT/grbm.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1
cd "$R"
echo "passes written under $OUT"
