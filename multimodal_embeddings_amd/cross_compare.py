"""north_star `cross_compare(vecs) -> sim`: the exact all-pairs cosine matrix.

The reference never forms this matrix; every "compare" is a ChromaDB HNSW
`collection.query` per vector (deprecated_package/cross_compare.py:119-123,
region_compare.py:165-170, weighted_region_clustering.py:79-84).  `cross_compare` is
the brute-force object those approximate kNN calls sample from, computed by the MFMA
cosine kernel (K9) over L2-normalised bf16 rows with f32 accumulation.
"""
from __future__ import annotations

import numpy as np

from ._lib import Engine, MmeError

_default_engine: Engine | None = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        import os

        _default_engine = Engine(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_engine


def to_unit_bf16(vecs, engine: Engine):
    """list-of-lists / ndarray / torch tensor [N,D] -> L2-normalised bf16 CUDA tensor."""
    t = engine.torch
    dev = t.device(f"cuda:{engine.device}")
    if isinstance(vecs, t.Tensor):
        x = vecs.to(dev)
        if x.dtype == t.bfloat16:
            return x.contiguous()  # engine outputs are unit rows already
        x = x.to(t.float32)
    else:
        if any(v is None for v in vecs):
            raise ValueError("vecs contains None holes; drop failed embeddings first (region_processor.py:132-138)")
        x = t.from_numpy(np.ascontiguousarray(np.asarray(vecs, dtype=np.float32))).to(dev)
    if x.ndim != 2:
        raise ValueError("vecs must be [N, D]")
    if x.shape[1] % 64 != 0:
        raise MmeError(f"D={x.shape[1]} must be a multiple of 64 for the MFMA cosine kernel")
    return engine.normalise_rows(x)


def cross_compare(vecs, other=None, *, engine: Engine | None = None, as_numpy: bool | None = None):
    """sim[i,j] = cos(vecs[i], other[j]) (other defaults to vecs).

    numpy / list input -> numpy float32 [N,M]; torch CUDA input -> CUDA tensor.
    """
    engine = engine or default_engine()
    t = engine.torch
    if as_numpy is None:
        as_numpy = not isinstance(vecs, t.Tensor)
    a = to_unit_bf16(vecs, engine)
    b = a if other is None else to_unit_bf16(other, engine)
    sim = engine.cosine(a, b)
    return sim.cpu().numpy() if as_numpy else sim
