#!/usr/bin/env python3
"""Compare / cluster stages at the C4 / C5 sizes on ONE GPU: N = 65536 regions in P = 512 pages
of 128, D = 768.  K9: a [N/8, N] row block of the cosine matrix (one rank's share of C4) and its HBM rate;
K10: the 512 x 512 page matrix; K11: clustering; K12: all ranked neighbour lists."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multimodal_embeddings_amd._lib import Engine
from multimodal_embeddings_amd.weighted_region_clustering import page_similarity_from_table


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def main():
    eng = Engine(0)
    P, per, d = 512, 128, 768
    n = P * per
    g = torch.Generator(device="cuda").manual_seed(5)
    centres = torch.randn(40, d, generator=g, device="cuda") * 1.5
    e16 = eng.normalise_rows(torch.randn(n, d, generator=g, device="cuda") + centres[torch.randint(0, 40, (n,), generator=g, device="cuda")])
    rows = n // 8
    sim = torch.empty((rows, n), dtype=torch.float32, device="cuda")
    wr, fl = rows * n * 4.0, 2.0 * rows * n * d
    for variant, name in ((1, "128 x 128 tiles, several workgroups per CU: one workgroup stores while others compute"), (3, "all stores in the epilogue (what mme_cosine launches)"), (4, "last row block deferred into the next tile's K loop")):
        eng.set_gemm_variant(variant)
        dt, _ = timed(lambda: eng.cosine(e16[:rows], e16, out=sim), reps=10)
        print(f"K9  cosine row block [{rows} x {n}] f32, variant {variant} ({name}): {dt*1e3:.3f} ms  write {wr/dt/1e12:.2f} TB/s = {wr/dt/8e12*100:.1f} % of 8 TB/s "
              f"(+{n*d*2/1e6:.0f} MB read)  {fl/dt/1e12:.0f} TFLOP/s", flush=True)
    eng.set_gemm_variant(0)
    sim16 = torch.empty((rows, n), dtype=torch.bfloat16, device="cuda")
    dt, _ = timed(lambda: eng.cosine_bf16(e16[:rows], e16, out=sim16), reps=10)
    print(f"K9  the same block with S as bf16 (mme_cosine_bf16): {dt*1e3:.3f} ms  write {wr/2/dt/1e12:.2f} TB/s  {fl/dt/1e12:.0f} TFLOP/s", flush=True)
    del sim16
    rng = np.random.default_rng(8)
    area = np.exp(rng.uniform(np.log(1e-2), np.log(20.0), n))
    valid = np.ones(n, np.uint8)
    offs = (np.arange(P + 1) * per).astype(np.int32)
    names = [f"{p:04d} synthetic page of the full-size set.png" for p in range(P)]
    dt, S = timed(lambda: page_similarity_from_table(e16, area, valid, offs, names, engine=eng))
    print(f"K10 page matrix P={P} (5120 query rows x {n}): {dt*1e3:.2f} ms  ({10*P*n*4/dt/1e12:.2f} TB/s of qsim)", flush=True)
    Sh = S.cpu().numpy()
    np.fill_diagonal(Sh, 1.0)
    dt, out = timed(lambda: eng.cluster_pages(Sh), reps=2)
    print(f"K11 clustering P={P}: {dt*1e3:.1f} ms (k={out[1]})", flush=True)
    group = torch.from_numpy((np.arange(n) // per).astype(np.int32)).cuda()
    dt, _ = timed(lambda: eng.neighbours(e16, group, fetch=30, top_n=10))
    print(f"K12 neighbour lists N={n}: {dt*1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
