#!/usr/bin/env python3
"""Where a K-tile of the 256x256 GEMM spends its cycles: barrier-to-barrier intervals from the stamped build.

Slot naming for wave 0 (group 0): D0 M0 D1 M1 D2 M2 D3 M3 = data / MFMA halves of the four phases; group 1
(wave 4) runs one barrier behind, so its interval k overlaps group 0's interval k+1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from multimodal_embeddings_amd._lib import Engine


def main():
    eng = Engine(0)
    for name, M, N, K in [("qkv", 806912, 2304, 768), ("fc1", 806912, 3072, 768), ("fc2", 806912, 768, 3072)]:
        st = eng.gemm_stamps(M, N, K).astype(np.float64)
        nk = st[:, :, 10]
        ok = nk[:, 0] > 0
        for w, label in ((0, "wave0 (group 0)"), (1, "wave4 (group 1)")):
            per = st[ok, w, :8] / nk[ok, w, None]
            iv = per.mean(axis=0)
            print(f"{name:4s} {label}: cycles/K-tile {iv.sum():7.0f} | intervals " + " ".join(f"{x:5.0f}" for x in iv)
                  + f" | wait {np.mean(st[ok, w, 8] / nk[ok, w]):5.0f} | between tiles, per tile: total {np.mean(st[ok, w, 9] / nk[ok, w]) * K / 64:6.0f}"
                  f" = sync+prologue issue {np.mean(st[ok, w, 11] / nk[ok, w]) * K / 64:5.0f} + epilogue body {np.mean(st[ok, w, 12] / nk[ok, w]) * K / 64:5.0f} + rest", flush=True)


        clk = st[ok, 0, 13] / np.maximum(st[ok, 0, 14], 1.0) * 0.1  # GHz: shader cycles per 10 ns tick
        print(f"{name:4s} in-kernel clock (s_memtime / s_memrealtime x 100 MHz, median over workgroups): {np.median(clk):.3f} GHz "
              f"(min {clk.min():.3f}, max {clk.max():.3f}); kernel {np.median(st[ok, 0, 14]) / 100:.0f} us", flush=True)


if __name__ == "__main__":
    main()
