#!/usr/bin/env python3
"""PCIe-inclusive throughput of RegionEmbedder.get_image_embeddings from host arrays, by group size (batch_size x 16
crops per device pass): where the pipeline's fill / drain and the smaller passes trade off."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from multimodal_embeddings_amd.embedder import RegionEmbedder
from multimodal_embeddings_amd.weights import synthetic_crops


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    base = synthetic_crops(4096, seed=0)
    arrays = [base[i % 4096] for i in range(n)]
    emb = RegionEmbedder()
    emb.get_image_embeddings(arrays[:8192], batch_size=256, as_array=True)
    for bs in (32, 64, 128, 256):
        for as_array in (True, False):
            emb.get_image_embeddings(arrays[: 32 * bs], batch_size=bs, as_array=as_array)
            t0 = time.perf_counter()
            emb.get_image_embeddings(arrays, batch_size=bs, as_array=as_array)
            dt = time.perf_counter() - t0
            print(f"group {16 * bs:5d} crops, {'ndarray' if as_array else 'float lists'}: {n / dt:8.0f} crops/s ({dt * 1e3:.0f} ms for {n})", flush=True)


if __name__ == "__main__":
    main()
