#!/usr/bin/env python3
"""Interleaved A/B of K1's patch emitter in ONE process (diagnostic build): the affine form (one fma per value, verified
bit-exact on the host) against the table form (MME_K1_TABLE=1), on the all-224 x 224 batch (C2) and on the bundled size
distribution (C3).  Kernel time from the library's own events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_library

use_diag_library()
import numpy as np
import torch

from multimodal_embeddings_amd._lib import Engine
from multimodal_embeddings_amd.weights import make_vit_weights, synthetic_crops


def main():
    n = 4096
    eng = Engine(0)
    eng.load_vit(make_vit_weights(seed=1))
    sets = {}
    pix = torch.from_numpy(synthetic_crops(n, seed=0).reshape(-1)).cuda()
    sets["C2 224x224"] = (torch.cat([pix, torch.zeros(16, dtype=torch.uint8, device="cuda")]), np.arange(n, dtype=np.int64) * 150528,
                          np.tile(np.array([[224, 224]], dtype=np.int32), (n, 1)))
    sizes = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bundled_crop_sizes_hw.npy"))
    hw = sizes[np.arange(n) % len(sizes)].astype(np.int32)
    nbytes = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
    offs = np.zeros(n, dtype=np.int64)
    offs[1:] = np.cumsum((nbytes[:-1] + 15) // 16 * 16)
    g = torch.Generator(device="cuda").manual_seed(0)
    sets["C3 bundled sizes"] = (torch.randint(0, 256, (int(offs[-1] + nbytes[-1]) + 16,), dtype=torch.uint8, device="cuda", generator=g), offs, hw)
    for name, (p, o, h) in sets.items():
        ref = None
        res = {0: [], 1: []}
        for r in range(5):
            for table in (0, 1):
                if table:
                    os.environ["MME_K1_TABLE"] = "1"
                else:
                    os.environ.pop("MME_K1_TABLE", None)
                eng.profile(True)
                for _ in range(5):
                    out = eng.preprocess(p, o, h)
                torch.cuda.synchronize()
                ms = eng.profile_read()["preprocess"][0] / 5
                eng.profile(False)
                if r:
                    res[table].append(ms)
                if ref is None:
                    ref = out.clone()
                else:
                    assert torch.equal(ref.view(torch.int16), out.view(torch.int16)), "forms differ"
        print(f"{name}: affine {min(res[0]):.3f} ms (med {sorted(res[0])[len(res[0]) // 2]:.3f})   table {min(res[1]):.3f} ms (med {sorted(res[1])[len(res[1]) // 2]:.3f})   bit-identical", flush=True)


if __name__ == "__main__":
    main()
