#!/usr/bin/env python3
"""Per-shape GEMM microbenchmark (random bf16 data), all kernel variants in ONE process.

    python tools/gemm_bench.py
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_embeddings_amd._lib import Engine

SHAPES = [  # (name, M, N, K, epilogue) at chunk = 4096 crops
    ("qkv", 806912, 2304, 768, 0),
    ("proj", 806912, 768, 768, 2),
    ("fc1", 806912, 3072, 768, 1),
    ("fc2", 806912, 768, 3072, 2),
]


def main():
    eng = Engine(0)
    rounds = 3
    for name, M, N, K, epi in SHAPES:
        for variant in (3, 4):
            ms = [eng.gemm_bench(M, N, K, epi, variant, iters=5)[0] for _ in range(rounds)]
            print(f"{name:5s} v{variant}: ms min {min(ms):.4f} med {sorted(ms)[len(ms)//2]:.4f}  TF/s best {2.0*M*N*K/min(ms)/1e9:.0f}", flush=True)


if __name__ == "__main__":
    main()
