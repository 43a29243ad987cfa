#!/usr/bin/env python3
"""Ablations of the fast tile-ViT attention kernel (diagnostic build; results invalid): which part of an interval costs what.
MME_TATTN_DEBUG bits: 1 no exponentials, 2 no P.V MFMAs, 4 no S^T MFMAs, 8 no barrier, 16 no K/V staging."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for dbg in [0, 1, 2, 4, 6, 7, 8, 16, 24, 31]:
    env = dict(os.environ, MME_TATTN_DEBUG=str(dbg), MME_ALLOW_LIB_OVERRIDE="1", MME_LIB_PATH=os.path.join(ROOT, "multimodal_embeddings_amd", "libmme_diag.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_tilevit.py"), "8"], env=env, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    print(f"dbg {dbg:2d}: {line[-1] if line else r.stderr[-400:]}", flush=True)
