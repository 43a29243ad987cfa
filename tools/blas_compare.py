#!/usr/bin/env python3
"""Reference point for the hand-written GEMM: torch.matmul (hipBLASLt / rocBLAS) on the same four
bf16 shapes (plain GEMM, no epilogue), same process, random data."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_embeddings_amd._lib import Engine

SHAPES = [("qkv", 806912, 2304, 768), ("proj", 806912, 768, 768), ("fc1", 806912, 3072, 768), ("fc2", 806912, 768, 3072)]


def main():
    eng = Engine(0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(2):
            torch.matmul(a, w.t(), out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            torch.matmul(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        del a, w, out
        mine = min(eng.gemm_bench(M, N, K, 0, 3, iters=5)[0] for _ in range(2))
        print(f"{name:5s} torch.matmul {ms:.3f} ms ({2.0 * M * N * K / ms / 1e9:.0f} TF/s)   gemm_bf16_tn_256r+bias {mine:.3f} ms ({2.0 * M * N * K / mine / 1e9:.0f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
