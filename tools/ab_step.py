#!/usr/bin/env python3
"""A/B of whole embed steps in ONE process, interleaved rounds (guide rule 24): GEMM variants x attention buffers.

    python tools/ab_step.py [--crops 4096] [--rounds 4]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_library

use_diag_library()  # the MME_* experiment switches below exist only in libmme_diag.so
import numpy as np
import torch

from multimodal_embeddings_amd._lib import Engine
from multimodal_embeddings_amd.weights import make_vit_weights, synthetic_crops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--arms", default="4:2:2:1,4:2:2:2", help="gemm_variant:attention_buffers:ln_mode:attention_form (1 exact, 2 fast):gemm_rb:attention_dma_share, comma separated")
    args = ap.parse_args()
    n = args.crops
    eng = Engine(0)
    eng.load_vit(make_vit_weights(seed=1))
    pix = torch.empty(n * 224 * 224 * 3 + 16, dtype=torch.uint8, device="cuda")
    pix[: n * 224 * 224 * 3] = torch.from_numpy(synthetic_crops(n, seed=0).reshape(-1)).cuda()
    offs = np.arange(n, dtype=np.int64) * (224 * 224 * 3)
    hw = np.tile(np.array([[224, 224]], dtype=np.int32), (n, 1))
    arms = [tuple(int(v) for v in a.split(":")) for a in args.arms.split(",")]
    times = {a: [] for a in arms}
    kern = {}
    ref = None
    for r in range(args.rounds + 1):
        for a in arms:
            eng.set_gemm_variant(a[0])
            os.environ["MME_ATTN_BUFS"] = str(a[1])
            eng.set_ln_fusion(a[2] if len(a) > 2 else 2)
            os.environ["MME_ATTN_PIPE"] = str(a[3] if len(a) > 3 else 2)
            os.environ["MME_GEMM_RB"] = str(a[4] if len(a) > 4 else 0)
            os.environ["MME_ZIGZAG"] = str(a[6]) if len(a) > 6 else "0"
            if len(a) > 5:
                os.environ["MME_ATTN_SHARE"] = str(a[5])
            else:
                os.environ.pop("MME_ATTN_SHARE", None)
            eng.profile(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                e32, _ = eng.embed(pix, offs, hw, 0)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            prof = eng.profile_read()
            eng.profile(False)
            if r:  # round 0 warms up
                times[a].append(dt * 1e3)
                kern[a] = {k: v[0] / 3 for k, v in prof.items() if v[1]}
            if ref is None:
                ref = e32.clone()
            elif not torch.equal(ref, e32) and r == 0:
                print(f"arm {a}: embeddings differ from arm {arms[0]} (max 1-cos {float((1.0 - (ref * e32).sum(dim=1)).max()):.3e})", flush=True)
    for a in arms:
        t = sorted(times[a])
        print(f"gemm variant {a[0]} attn bufs {a[1]} ln mode {a[2] if len(a) > 2 else 2} attn pipe {a[3] if len(a) > 3 else 1} gemm rb {a[4] if len(a) > 4 else 0} attn share {a[5] if len(a) > 5 else 'default'} zigzag {a[6] if len(a) > 6 else 0}: ms/step min {t[0]:.2f} med {t[len(t) // 2]:.2f} | "
              + " ".join(f"{k} {v:.2f}" for k, v in kern[a].items()), flush=True)


if __name__ == "__main__":
    main()
