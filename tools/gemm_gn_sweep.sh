#!/bin/bash
# column-group width sweep of the GEMM tile order (MME_GEMM_GN is read once per process): one gemm_bench run per value
O=${1:-gpurun_out/gn}; mkdir -p $O
for gn in 0 3 4 5 6 9 12; do
  MME_GEMM_GN=$gn python3 tools/gemm_bench.py 2>/dev/null | grep " v4" | sed "s/^/GN=$gn /" | tee -a $O/sweep.log
done
