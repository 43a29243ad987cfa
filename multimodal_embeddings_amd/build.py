"""Build libmme.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m multimodal_embeddings_amd.build [--force] [--diag]

`--diag` builds libmme_diag.so instead (-DMME_DIAG: the only build that reads the experiment switches of DESIGN.md
4.5 from the environment; measurement tools select it with MME_LIB_PATH).

hipcc cross-compiles gfx950 code objects without a GPU, so this also is the
"does it build" check of __graft_entry__.build().  The .so stays inside the
package directory (git-ignored, but it travels to the GPU box with the tree).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libmme.so")
SOURCES = ["gemm.hip", "gemm256r.hip", "rowops.hip", "attention.hip", "preprocess.hip", "page_reduce.hip", "cluster.hip", "neighbours.hip", "launch_state.hip", "comm.hip", "nms.hip", "attention_tiles.hip", "tilevit.hip", "capi.hip", "capi_tilevit.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_epilogue.h", "ctx.h", os.path.join("..", "..", "include", "mme.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def kernel_source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/mme.h), names included: what a counter record
    (profiles/current.json) is valid for.  bench.py compares it with the record's and marks the record stale."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(PKG, "..", "include", "mme.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    hipcc = _hipcc()
    OBJ = os.path.join(CSRC, "_obj_diag" if diag else "_obj")
    LIB = os.path.join(PKG, "libmme_diag.so" if diag else "libmme.so")
    FLAGS = globals()["FLAGS"] + (["-DMME_DIAG"] if diag else [])
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv)
    print(path)
