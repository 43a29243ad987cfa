#!/usr/bin/env python3
"""Which part of the K9 block [8192 x 65536] f32 costs what (diagnostic library, MME_GEMM_DEBUG; results invalid in modes 1 / 2):
0 the real launch, 1 no K-loop operand loads (MFMAs + the f32 epilogue stores alone), 2 every tile reads tile 0 (operands always
hit the L2).  One child process per mode, variant 3 (what mme_cosine launches)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, torch
sys.path.insert(0, %r)
from multimodal_embeddings_amd._lib import Engine
eng = Engine(0)
n, d = 65536, 768
g = torch.Generator(device="cuda").manual_seed(5)
e16 = eng.normalise_rows(torch.randn(n, d, generator=g, device="cuda"))
rows = n // 8
sim = torch.empty((rows, n), dtype=torch.float32, device="cuda")
eng.set_gemm_variant(3)
for _ in range(3): eng.cosine(e16[:rows], e16, out=sim)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): eng.cosine(e16[:rows], e16, out=sim)
torch.cuda.synchronize()
print("%%.3f" %% ((time.perf_counter() - t0) / 30 * 1e3))
""" % ROOT
for mode, what in ((0, "real launch"), (1, "no operand loads: MFMAs + stores"), (2, "operands always in L2"), (0, "real launch again")):
    env = dict(os.environ, MME_GEMM_DEBUG=str(mode), MME_ALLOW_LIB_OVERRIDE="1", MME_LIB_PATH=os.path.join(ROOT, "multimodal_embeddings_amd", "libmme_diag.so"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    ms = [ln for ln in r.stdout.splitlines() if ln.replace(".", "").isdigit()]
    print(f"MME_GEMM_DEBUG={mode} ({what}): {ms[-1] if ms else r.stderr[-300:]} ms", flush=True)
