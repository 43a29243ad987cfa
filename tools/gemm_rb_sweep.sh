#!/bin/bash
# row-panel blocking sweep of the GEMM tile order (MME_GEMM_RB, with the column-group width MME_GEMM_GN): one gemm_bench run per pair.
# rb was measured at 8 / 16 / 32 only; the L2-sized end (2..4 row panels of 256 rows = 0.8..1.6 MB of A per K = 768 panel) here.
O=${1:-gpurun_out/rb}; mkdir -p $O
export MME_ALLOW_LIB_OVERRIDE=1 MME_LIB_PATH=$PWD/multimodal_embeddings_amd/libmme_diag.so; test -f $MME_LIB_PATH || { echo "build libmme_diag.so first"; exit 1; }
for cfg in "0 0" "2 0" "3 0" "4 0" "6 0" "2 12" "4 12" "0 12" "0 0"; do
  set -- $cfg
  MME_GEMM_RB=$1 MME_GEMM_GN=$2 python3 tools/gemm_bench.py 2>/dev/null | grep " v4" | sed "s/^/RB=$1 GN=$2 /" | tee -a $O/sweep.log
done
