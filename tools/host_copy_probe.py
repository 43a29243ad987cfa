#!/usr/bin/env python3
"""Host -> device staging rates on this box: what bounds the PCIe-inclusive paths (process_regions, get_image_embeddings)."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

nb = 128 << 20
src = np.random.default_rng(0).integers(0, 256, nb, dtype=np.uint8)
pin = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
hv = pin.numpy()
dev = torch.empty(nb, dtype=torch.uint8, device="cuda")
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for workers in (1, 2, 4, 8, 12, 16):
    pool = ThreadPoolExecutor(max_workers=workers)
    cuts = [nb * w // workers for w in range(workers + 1)]
    def run():
        list(pool.map(lambda ab: hv.__setitem__(slice(ab[0], ab[1]), src[ab[0]:ab[1]]), zip(cuts[:-1], cuts[1:])))
    run()
    t = time.perf_counter()
    for _ in range(5):
        run()
    dt = (time.perf_counter() - t) / 5
    print(f"memcpy pageable -> pinned, {workers:2d} threads: {nb / dt / 1e9:6.1f} GB/s")
    pool.shutdown()
torch.cuda.synchronize()
for _ in range(2):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D from pinned: {nb * 10 / (time.perf_counter() - t) / 1e9:6.1f} GB/s")
pg = torch.from_numpy(src)
dev.copy_(pg)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    dev.copy_(pg)
torch.cuda.synchronize()
print(f"H2D from pageable (driver staging): {nb * 5 / (time.perf_counter() - t) / 1e9:6.1f} GB/s")
# strided crop copy: 400 x 1200 px windows of a 5000 x 4000 page into the pinned buffer
page = src[: 5000 * 4000 * 3].reshape(5000, 4000, 3)
t = time.perf_counter()
o = 0
for k in range(60):
    y, x = (k * 70) % 4500, (k * 40) % 2700
    c = page[y : y + 400, x : x + 1200]
    hv[o : o + c.size].reshape(400, 1200, 3)[...] = c
    o += c.size
dt = time.perf_counter() - t
print(f"strided crop copies into pinned, 1 thread: {o / dt / 1e9:6.1f} GB/s ({o / 1e6:.0f} MB)")

# ---- the same copies WHILE the encoder runs (what the pipelined host paths actually see)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_embeddings_amd._lib import Engine  # noqa: E402
from multimodal_embeddings_amd.weights import make_vit_weights  # noqa: E402

eng = Engine(0)
eng.load_vit(make_vit_weights(seed=1))
n = 4096
pix = torch.randint(0, 256, (n * 150528 + 16,), dtype=torch.uint8, device="cuda")
offs = np.arange(n, dtype=np.int64) * 150528
hw = np.tile(np.array([[224, 224]], dtype=np.int32), (n, 1))
eng.embed(pix, offs, hw)
torch.cuda.synchronize()
side = torch.cuda.Stream()
for label, srct in (("pinned, async", pin), ("pageable, blocking", pg)):
    t0 = time.perf_counter()
    for _ in range(4):
        eng.embed(pix, offs, hw)  # ~0.57 s of device work queued on the main stream
    t_enq = time.perf_counter() - t0
    t = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(20):
            dev.copy_(srct, non_blocking=True)
        side.synchronize()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"H2D beside the encoder, {label}: {nb * 20 / dt / 1e9:6.1f} GB/s (enqueue of 4 passes {t_enq * 1e3:.0f} ms, copies {dt * 1e3:.0f} ms, all done {t_all * 1e3:.0f} ms)")
