#!/usr/bin/env python3
"""What the residual read costs the o_proj / fc2 GEMMs: the same shape with the plain bias epilogue and with bias + residual
(in place), three rounds in one process."""
import sys, os
sys.path.insert(0, os.getcwd())
from multimodal_embeddings_amd._lib import Engine
eng = Engine(0)
M = 806912
for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072)):
    for rnd in range(3):
        r = []
        for epi, label in ((0, "bias"), (2, "bias+residual")):
            try:
                ms = min(eng.gemm_bench(M, N, K, epi, 4, iters=5)[0] for _ in range(2))
            except Exception as e:
                ms = float("nan")
            r.append(f"{label} {ms:.3f} ms")
        print(name, "round", rnd, " | ".join(r), flush=True)
