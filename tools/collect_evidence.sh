#!/bin/bash
# Round evidence beyond the headline command (run on the GPU box from the repo root):
#   bash tools/collect_evidence.sh gpurun_out/ev_r2
# kernel-trace summaries of the C3 / C5 / K12 / small-batch / tile-ViT tools, the bench lines of the three bench
# configurations, and the stamped diagnostic builds.  tools/copy_evidence.py then copies the summaries into profiles/.
set -e
OUT=$(realpath -m "$1")
R=$(pwd)
export TMPDIR=/tmp
rm -rf "$OUT"  # gpurun merges into an existing gpurun_out/: stale pass files of an earlier call would be folded in twice
mkdir -p "$OUT"
cd /tmp
for t in bench_c3 bench_c5 bench_neighbours bench_small bench_tilevit bench_regions; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$t" -- python3 "$R/tools/$t.py" > "$OUT/$t.log" 2>&1
    echo "$t done"
done
cd "$R"
python3 bench.py --steps 10 --warmup 2 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python3 bench.py --config c3 --steps 5 --warmup 2 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
python3 bench.py --config c4 --steps 4 --warmup 1 --headline-only > "$OUT/bench_c4_1gpu.json" 2> "$OUT/bench_c4_1gpu.err"
python3 bench.py --config c5 --steps 2 --warmup 1 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
python3 bench.py --config tilevit --steps 3 --warmup 1 > "$OUT/bench_tilevit.json" 2> "$OUT/bench_tilevit.err"
# the N > 1 code path on real RCCL at world size 1 (process group forced): C4 share and the C5 chain
python3 bench.py --config c4 --steps 3 --warmup 1 --headline-only --force-dist > "$OUT/bench_c4_nccl_world1.json" 2> "$OUT/bench_c4_nccl_world1.err"
python3 bench.py --config c5 --crops 8192 --steps 2 --warmup 1 --no-cpu-baseline --force-dist > "$OUT/bench_c5_nccl_world1.json" 2> "$OUT/bench_c5_nccl_world1.err"
MME_DIST_BACKEND=gloo python3 bench.py --gpus 2 --crops 1024 --steps 3 --warmup 1 > "$OUT/bench_gloo_2ranks_1gpu.json" 2> "$OUT/bench_gloo_2ranks_1gpu.err"
MME_DIST_BACKEND=gloo python3 bench.py --gpus 2 --config c5 --crops 2048 --steps 2 --warmup 1 > "$OUT/bench_gloo_2ranks_1gpu_c5.json" 2> "$OUT/bench_gloo_2ranks_1gpu_c5.err"
python3 tools/bench_from_host.py > "$OUT/bench_from_host.log" 2>&1
python3 tools/gemm_stamps.py > "$OUT/gemm_stamps.log" 2>&1
# the stamped attention build and the A/B tools read their switches from the diagnostic library only
python3 tools/attn_stamps.py > "$OUT/attn_stamps.log" 2>&1
[ -n "$EVIDENCE_SKIP_ABLATIONS" ] || python3 tools/tattn_ablate.py > "$OUT/tattn_ablations.log" 2>&1
python3 tools/host_copy_probe.py > "$OUT/host_copy_probe.log" 2>&1
echo "evidence written under $OUT"
