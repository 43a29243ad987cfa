#!/usr/bin/env python3
"""Per-shape GEMM microbenchmark (random bf16 data), all kernel variants in ONE process.

    python tools/gemm_bench.py [--stamps]
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from multimodal_embeddings_amd._lib import Engine

SHAPES = [  # (name, M, N, K, epilogue) at chunk = 1024 crops
    ("qkv", 201728, 2304, 768, 0),
    ("proj", 201728, 768, 768, 2),
    ("fc1", 201728, 3072, 768, 1),
    ("fc2", 201728, 768, 3072, 2),
]


def main():
    eng = Engine(0)
    stamps = "--stamps" in sys.argv
    rounds = 3
    for name, M, N, K, epi in SHAPES:
        for variant in (1, 2, 3, 4):
            res = [eng.gemm_bench(M, N, K, epi, variant, iters=5) for _ in range(rounds)]
            ms = [r[0] for r in res]
            print(f"{name:5s} v{variant}: ms min {min(ms):.4f} med {sorted(ms)[len(ms)//2]:.4f}  TF/s best {2.0*M*N*K/min(ms)/1e9:.0f}", flush=True)
        if stamps:
            ms, tf, st = eng.gemm_bench(M, N, K, epi, 3, iters=2, stamps=True)
            st = st[st[:, 5] > 0].astype(np.float64)
            tot, lds, vm, bar, epi_c, nkt = (st[:, i] for i in range(6))
            per = tot / nkt
            print(f"      stamps v3: waves {len(st)}  cycles/K-tile {per.mean():.0f}  lds-drain {np.mean(lds/nkt):.0f}  dma-wait {np.mean(vm/nkt):.0f}"
                  f"  barrier {np.mean(bar/nkt):.0f}  epilogue/K-tile {np.mean(epi_c/nkt):.0f}  (K-tiles/wave {nkt.mean():.0f})", flush=True)


if __name__ == "__main__":
    main()
