#!/usr/bin/env python3
"""Interleaved A/B of the K9 cosine block [8192 x 65536] f32: GEMM variant 3 (all stores in the epilogue) against variant 4
(last row block deferred into the next tile's K loop), 20 launches per arm, four rounds, one process."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from multimodal_embeddings_amd._lib import Engine
eng = Engine(0)
n, d = 65536, 768
g = torch.Generator(device="cuda").manual_seed(5)
e16 = eng.normalise_rows(torch.randn(n, d, generator=g, device="cuda"))
rows = n // 8
sim = torch.empty((rows, n), dtype=torch.float32, device="cuda")
def t(v, reps=20):
    eng.set_gemm_variant(v)
    eng.cosine(e16[:rows], e16, out=sim); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): eng.cosine(e16[:rows], e16, out=sim)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for r in range(4):
    print("round", r, " v3 %.3f ms  v4 %.3f ms  v3 %.3f  v4 %.3f" % (t(3), t(4), t(3), t(4)), flush=True)
