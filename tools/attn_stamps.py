#!/usr/bin/env python3
"""Where a head iteration of the attention kernel (K5) spends its cycles, per wave, from the stamped build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_library

use_diag_library()  # the MME_* experiment switches below exist only in libmme_diag.so
import numpy as np

from multimodal_embeddings_amd._lib import Engine

NAMES = ["wait own", "barrier", "issue next", "S^T", "softmax", "P.V", "handover+stores"]


def main():
    eng = Engine(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    shares = [v for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else [""])]
    for pipe, share in [(int(v), sh) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "2"]) for sh in shares]:
        os.environ["MME_ATTN_PIPE"] = str(pipe)
        if share != "":
            os.environ["MME_ATTN_SHARE"] = share
            print(f"=== K/V pieces requested by each computing wave: {share}")
        for dbg in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]:
            os.environ["MME_ATTN_DEBUG"] = str(dbg)
            print(f"--- kernel form {pipe} (1 exact row maximum, 2 fast / reference point from key tile 0); MME_ATTN_DEBUG={dbg} "
                  "(exact stamped build only: 1 no K re-reads, 2 no K/V requests after head 0, 4 no max, 8 no exp)")
            report(eng, B)


def report(eng, B):
    ms, st = eng.attention_stamps(B, iters=10)
    st = st.astype(np.float64)
    heads = st[:, :, 7]
    per = st[:, :, :7] / np.maximum(heads[:, :, None], 1)
    print(f"attention B={B}: {ms:.3f} ms per launch = {ms * 1e3 / (B / 256.0) / 12:.2f} us per head per CU-slot; bytes {B * 197 * 3072 * 2 / ms / 1e9:.2f} TB/s")
    print("cycles per head (mean over workgroups):   " + "  ".join(f"{n:>15s}" for n in NAMES) + "    total")
    clk = st[:, 7, 5] / np.maximum(st[:, 7, 6], 1.0) * 0.1
    print(f"in-kernel clock of the stamped build (s_memtime / s_memrealtime x 100 MHz, median over workgroups): {np.median(clk):.3f} GHz")
    per[:, 7, 5:7] = 0
    for w in range(8):
        m = per[:, w].mean(axis=0)
        print(f"wave {w}{' (staging)' if w == 7 else '          '}                     " + "  ".join(f"{v:15.0f}" for v in m) + f"  {m.sum():7.0f}")


if __name__ == "__main__":
    main()
