#!/usr/bin/env python3
"""Config C3: 4096 variable-size crops (the size distribution of the reference's 1862 bundled
region crops, cycled; pixels synthetic because the crops themselves cannot travel) through the
on-GPU resize / normalise / patchify (K1) and the embedder.  Prints K1 time, algorithmic bytes,
GB/s and whole-path crops/s, and checks a sample against the oracle bit-for-bit (patches).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multimodal_embeddings_amd._lib import Engine
from multimodal_embeddings_amd.weights import make_vit_weights, round_to_bf16


def main():
    n = 4096
    sizes = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bundled_crop_sizes_hw.npy"))
    hw = sizes[np.arange(n) % len(sizes)].astype(np.int32)
    nbytes = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
    offs = np.zeros(n, dtype=np.int64)
    offs[1:] = np.cumsum((nbytes[:-1] + 15) // 16 * 16)
    total = int(offs[-1] + nbytes[-1]) + 16
    g = torch.Generator(device="cuda").manual_seed(0)
    pix = torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda", generator=g)
    eng = Engine(0)
    eng.load_vit(make_vit_weights(seed=1))
    eng.profile(True)
    for _ in range(2):
        patches = eng.preprocess(pix, offs, hw)
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        patches = eng.preprocess(pix, offs, hw)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    prof = eng.profile_read()
    k1_ms = prof["preprocess"][0] / reps
    alg = float(nbytes.sum()) + n * 301056.0
    print(f"C3 K1: {n} crops, {nbytes.sum()/1e6:.1f} MB of pixels, algorithmic {alg/1e9:.3f} GB; kernels {k1_ms:.3f} ms "
          f"({alg/k1_ms/1e6:.0f} GB/s), wall incl. host crop table + H2D {wall*1e3:.2f} ms")
    # whole path
    for _ in range(2):
        eng.embed(pix, offs, hw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        e32, _ = eng.embed(pix, offs, hw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"C3 embed: {n / dt:.0f} crops/s ({dt*1e3:.1f} ms)")
    # sample parity (bit-exact patches)
    from oracle import preprocess as opre

    host = pix.cpu().numpy()
    got = patches.float().cpu().numpy().reshape(n, 196, 768)
    for i in (0, 1, 17, 500, 1861, 4095):
        h, w = hw[i]
        a = host[offs[i] : offs[i] + nbytes[i]].reshape(h, w, 3)
        assert np.array_equal(got[i], round_to_bf16(opre.preprocess_to_patches(a))), i
    print("C3 sample parity: bit-exact")


if __name__ == "__main__":
    main()
