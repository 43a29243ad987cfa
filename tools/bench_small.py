#!/usr/bin/env python3
"""Latency of small batches (the reference embeds <= 48 regions per call, region_processor.py:124-129):
crops resident in HBM, 224x224, per-call wall time with a device sync (what a caller of
get_image_embeddings waits for, minus decode and H2D)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_embeddings_amd.embedder import RegionEmbedder
from multimodal_embeddings_amd.weights import synthetic_crops


def main():
    emb = RegionEmbedder()
    variants = [int(v) for v in sys.argv[1:]] or [0]  # GEMM variants to compare (0 = the library's choice by problem size)
    for variant in variants:
      emb.engine.set_gemm_variant(variant)
      if len(variants) > 1:
          print(f"--- GEMM variant {variant}", flush=True)
      sizes = [int(v) for v in os.environ.get("BENCH_SMALL_N", "1,16,32,48,64,96,128,192,256,1024").split(",")]
      for n in sizes:
        crops = torch.from_numpy(synthetic_crops(n, seed=0)).cuda()
        for _ in range(3):
            emb.embed_uniform(crops)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            emb.embed_uniform(crops)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"n={n:5d}: {dt*1e3:8.3f} ms per call  ({n/dt:9.0f} crops/s)", flush=True)


if __name__ == "__main__":
    main()
