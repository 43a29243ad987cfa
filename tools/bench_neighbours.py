#!/usr/bin/env python3
"""K12 neighbour service at the C4 problem size on one GPU: N = 65536 regions, D = 768, 512 pages of 128.

    python tools/bench_neighbours.py [N]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multimodal_embeddings_amd._lib import Engine


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    d = 768
    eng = Engine(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    centres = torch.randn(64, d, generator=g, device="cuda") * 1.5
    x = torch.randn(n, d, generator=g, device="cuda") + centres[torch.randint(0, 64, (n,), generator=g, device="cuda")]
    e16 = eng.normalise_rows(x)
    group = torch.from_numpy((np.arange(n) // 128).astype(np.int32)).cuda()
    eng.neighbours(e16, group, fetch=30, top_n=10)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.neighbours(e16, group, fetch=30, top_n=10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flop = 2.0 * n * n * d
    bytes_min = n * n * 4.0 * 2  # the cosine block written once and read once (chunk-wise, never whole)
    print(f"neighbours N={n}: {dt * 1e3:.2f} ms  ({n / dt:.0f} query rows/s, {flop / dt / 1e12:.0f} TFLOP/s GEMM-equivalent, "
          f"{bytes_min / dt / 1e12:.2f} TB/s of cosine-block traffic)")


if __name__ == "__main__":
    main()
