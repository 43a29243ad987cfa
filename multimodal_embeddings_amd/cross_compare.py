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


def image_neighbours(vecs, filenames, top_n=None, *, score="cosine", engine: Engine | None = None):
    """Per image: its `top_n` most similar images with a different filename prefix.

    Mirrors the query + filter loop of `create_cross_comparison`
    (deprecated_package/cross_compare.py:109-235): fetch min(5*top_n, 100) nearest images (:117),
    drop the source (:176) and every candidate whose filename starts with the SOURCE's first
    max(1, int(len*0.2)) characters (:109-110, :200-206), keep the first top_n (:234).  The prefix
    length depends on the source, so this rule is applied on the host to the kernel's raw ranked
    list (K12 with keep_self).  `score="reference_distance"` reports the store's cosine distance,
    which is what :210 prints as "similarity".
    Returns a list (one per image) of [{"index", "filename", "prefix", "score"}, ...].
    """
    from . import config

    top_n = config.CROSS_COMPARE_TOP_N if top_n is None else top_n
    engine = engine or default_engine()
    emb = to_unit_bf16(vecs, engine)
    if emb.shape[0] != len(filenames):
        raise ValueError("one filename per vector required")
    fetch = min(top_n * 5, 100)
    idx, sim = engine.neighbours(emb, None, fetch=fetch, top_n=fetch, keep_self=True)
    idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
    out = []
    for r, name in enumerate(filenames):
        plen = max(1, int(len(name) * 0.2))
        src_prefix = name[:plen]
        lst = []
        for c, s in zip(idx[r], sim[r]):
            if c < 0:
                break
            if c == r:
                continue
            cand = filenames[c]
            prefix = cand[:plen] if len(cand) >= plen else cand
            if prefix == src_prefix:
                continue
            lst.append({"index": int(c), "filename": cand, "prefix": prefix,
                        "score": float(s) if score == "cosine" else float(1.0 - s)})
            if len(lst) >= top_n:
                break
        out.append(lst)
    return out
