#!/bin/bash
# column-group width sweep of the GEMM tile order (MME_GEMM_GN is read once per process): one gemm_bench run per value
O=${1:-gpurun_out/gn}; mkdir -p $O
# the switch exists only in the diagnostic build (python -m multimodal_embeddings_amd.build --diag)
export MME_ALLOW_LIB_OVERRIDE=1 MME_LIB_PATH=$PWD/multimodal_embeddings_amd/libmme_diag.so; test -f $MME_LIB_PATH || { echo "build libmme_diag.so first"; exit 1; }
for gn in 0 3 4 5 6 9 12; do
  MME_GEMM_GN=$gn python3 tools/gemm_bench.py 2>/dev/null | grep " v4" | sed "s/^/GN=$gn /" | tee -a $O/sweep.log
done
