#!/usr/bin/env python3
"""Tile-ViT option: time the full 32 + 8 layer Mllama vision tower on four-tile images."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multimodal_embeddings_amd._lib import Engine
from multimodal_embeddings_amd.embedder import RegionEmbedder
from multimodal_embeddings_amd.weights import make_tile_vit_weights

FLOP_PER_IMAGE = 40 * (4 * 6432 * 6432 * 1280 + 2 * 6432 * (1280 * 3840 + 1280 * 1280 + 2 * 1280 * 5120)) + 2 * 6400 * 588 * 1280


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    eng = Engine(0)
    t0 = time.time()
    w = make_tile_vit_weights(2)
    t1 = time.time()
    eng.load_tile_vit(w)
    print(f"weights: generate {t1 - t0:.1f} s, fold + upload {time.time() - t1:.1f} s", flush=True)
    rng = np.random.default_rng(0)
    arrays = [rng.integers(0, 256, (1000 + 7 * k, 1100, 3), dtype=np.uint8) for k in range(n)]
    emb = RegionEmbedder(engine=eng)
    pix, offs, hw = emb.pack(arrays)
    pv, ids, _, nt = eng.preprocess_tiles(pix, offs, hw, 560, 4)
    eng.set_chunk(n)
    eng.tile_vit_forward(pv, ids, nt)
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        eng.tile_vit_forward(pv, ids, nt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = eng.profile_read()
    print(f"{n} images x 4 tiles: {dt * 1e3:.1f} ms per pass = {dt * 1e3 / n:.2f} ms per image = {n / dt:.1f} images/s; "
          f"{FLOP_PER_IMAGE * n / dt / 1e12:.0f} TFLOP/s ({FLOP_PER_IMAGE / 1e12:.2f} TFLOP per image)")
    print({k: round(v[0] / reps, 2) for k, v in prof.items() if v[1]})


if __name__ == "__main__":
    main()
