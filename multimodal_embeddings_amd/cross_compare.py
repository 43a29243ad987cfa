"""north_star `cross_compare(vecs) -> sim`: the exact all-pairs cosine matrix.

The reference never forms this matrix; every "compare" is a ChromaDB HNSW
`collection.query` per vector (deprecated_package/cross_compare.py:119-123,
region_compare.py:165-170, weighted_region_clustering.py:79-84).  `cross_compare` is
the brute-force object those approximate kNN calls sample from, computed by the MFMA
cosine kernel (K9) over L2-normalised bf16 rows with f32 accumulation.
"""
from __future__ import annotations

import logging

import numpy as np

from ._lib import Engine, MmeError

logger = logging.getLogger(__name__)
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


def _embedding_from_db(collection, image_id):
    """db_operations.py:65-86 `get_embedding_from_db`: (embedding, success)."""
    try:
        got = collection.get(ids=[image_id], include=["embeddings"])
        if len(got["ids"]) > 0 and got.get("embeddings") is not None and len(got["embeddings"]) > 0 \
                and got["embeddings"][0] is not None and len(got["embeddings"][0]) > 0:
            return got["embeddings"][0], True
    except Exception as e:  # noqa: BLE001
        logger.error(f"Error getting embedding for {image_id}: {e}")
    return None, False


def process_image(embedder, collection, image_path):
    """image_processor.py:19-113 without the orientation step (out of scope): embed one whole image and add /
    update its row -- id `image_<filename>`, metadata {image_name, image_path (absolute), processed_time, is_region: False}."""
    import datetime
    import os

    name = os.path.basename(image_path)
    image_id = f"image_{name}"
    try:
        existing = collection.get(ids=[image_id], include=["embeddings", "metadatas", "documents"])
        if _embedding_from_db(collection, image_id)[1]:
            return True
        rows = embedder.get_image_embeddings([image_path], is_query=False)
        if not rows or rows[0] is None:
            logger.error(f"Failed to generate embedding for {name}")
            return False
        meta = {"image_name": name, "image_path": os.path.abspath(image_path), "processed_time": str(datetime.datetime.now()), "is_region": False}
        call = collection.update if len(existing["ids"]) > 0 else collection.add
        call(ids=[image_id], embeddings=[rows[0]], metadatas=[meta], documents=[f"Image: {name}"])
        return True
    except Exception as e:  # noqa: BLE001  (image_processor.py:111-113)
        logger.error(f"Error embedding/storing {name}: {e}")
        return False


def create_cross_comparison(embedder, collection, image_paths, top_n=None, output_path=None, *, score="reference_distance",
                            require_existing_paths=True, query_batch=64):
    """Same entry point and selection as `create_cross_comparison(embedder, collection, image_paths, top_n)`
    (deprecated_package/cross_compare.py:19-283); the report is one JSON document instead of HTML pages.

    Per image path, in order: id `image_<filename>` (:78); its vector from the collection (:92), re-embedded through
    `embedder` and stored when missing (:94-106), skipped when that fails; ONE unfiltered `collection.query` for
    min(5 * top_n, 100) neighbours (:117-123); then the reference's walk over the ranked list: drop the source (:148),
    rows without dict metadata (:152-160), rows whose `image_path` is empty or not on disk (:163-166: region rows have
    none), rows whose filename shares the source's first max(1, int(len * 0.2)) characters (:109-110, :168-174), keep the
    first `top_n` (:231-232).  `score="reference_distance"` reports the store's distance, which is what the reference
    prints as "Similarity score" (:177-183); "cosine" reports 1 - d for a cosine store.

    The queries of `query_batch` images go to the store in one call (chroma's `query` takes several query vectors; the
    RegionCollection ranks them in one K12 launch).  Returns [{"image", "id", "prefix", "similar": [{"id", "filename",
    "prefix", "score"}]}] for the images that got a page; writes {"top_n", "images"} to `output_path` when given."""
    import json
    import os

    from . import config

    top_n = config.CROSS_COMPARE_TOP_N if top_n is None else top_n
    logger.info(f"Starting cross-comparison (top {top_n} similar images, excluding files with similar prefixes)")
    todo = []  # (path, filename, id, vector)
    for image_path in image_paths:
        name = os.path.basename(image_path)
        image_id = f"image_{name}"
        vec, ok = _embedding_from_db(collection, image_id)
        if not ok:
            logger.warning(f"No valid embedding found for {name} in DB.")
            logger.info(f"Attempting to regenerate embedding for {name}...")
            if embedder is not None and process_image(embedder, collection, image_path):
                vec, ok = _embedding_from_db(collection, image_id)
                if not ok:
                    logger.error(f"Failed to regenerate embedding for {name}. Skipping.")
                    continue
            else:
                logger.error(f"Failed to process {name}. Skipping.")
                continue
        todo.append((image_path, name, image_id, vec))
    query_size = min(top_n * 5, 100)
    pages = []
    for b0 in range(0, len(todo), max(1, int(query_batch))):
        part = todo[b0: b0 + max(1, int(query_batch))]
        try:
            res = collection.query(query_embeddings=[v for _, _, _, v in part], n_results=query_size,
                                   include=["metadatas", "documents", "distances"])
        except Exception as e:  # noqa: BLE001  (:268-270)
            for _, name, _, _ in part:
                logger.error(f"Error in cross-comparison for {name}: {e}")
            continue
        for qi, (image_path, name, image_id, _) in enumerate(part):
            ids = res["ids"][qi] if res and "ids" in res and len(res["ids"]) > qi else []
            if not ids:
                logger.warning(f"No results found for {name}")  # :126-128
                continue
            plen = max(1, int(len(name) * 0.2))
            src_prefix = name[:plen]
            metas = res["metadatas"][qi] if res.get("metadatas") else []
            dists = res["distances"][qi] if res.get("distances") else []
            similar = []
            for i, rid in enumerate(ids):
                if rid == image_id:
                    continue
                meta = metas[i] if len(metas) > i else None
                if not meta or not isinstance(meta, dict):
                    continue
                cand_path = meta.get("image_path", "")
                if not cand_path or (require_existing_paths and not os.path.exists(cand_path)):
                    continue
                cand = os.path.basename(cand_path)
                prefix = cand[:plen] if len(cand) >= plen else cand
                if prefix == src_prefix:
                    continue
                d = float(dists[i]) if len(dists) > i and dists[i] is not None else None
                similar.append({"id": rid, "filename": cand, "prefix": prefix,
                                "score": d if score == "reference_distance" or d is None else 1.0 - d})
                if len(similar) >= top_n:
                    break
            pages.append({"image": name, "id": image_id, "prefix": src_prefix, "similar": similar})
    if output_path:
        os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
        with open(output_path, "w") as fh:
            json.dump({"top_n": top_n, "images": pages}, fh, indent=1)
    logger.info(f"Cross-comparison: {len(pages)}/{len(image_paths)} images reported")
    return pages
