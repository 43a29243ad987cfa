"""Ranked "similar regions from other pages" lists -- the data of the reference's region report.

Mirrors `create_region_cross_comparison` (deprecated_package/region_compare.py:25-406) without
its HTML / cv2 output (emit JSON instead): the per-region store query
(:165-170) plus the filter loop (:238-353) run as ONE pass of the K12 kernel
(`mme_neighbours`: MFMA cosine block + streaming top-k) over every region at once.
"""
from __future__ import annotations

import json
import logging
import os

import numpy as np

from . import config
from ._lib import Engine
from .cross_compare import default_engine, to_unit_bf16

logger = logging.getLogger(__name__)


def _box_of(meta):
    """region_compare.py:133-146: box from `box_str` or the four box_* keys, else None."""
    if meta.get("box_str"):
        return [float(x) for x in meta["box_str"].split(",")]
    keys = ["box_x_min", "box_y_min", "box_x_max", "box_y_max"]
    if all(k in meta for k in keys):
        return [meta[k] for k in keys]
    return None


def region_neighbours(collection, top_n=config.REGION_COMPARE_TOP_N, *, score="cosine",
                      threshold=config.REGION_SIMILARITY_THRESHOLD, weight_by_area=config.WEIGHT_BY_AREA,
                      engine: Engine | None = None, rows: tuple[int, int] | None = None):
    """For every region of the collection: its `top_n` most similar regions from other pages.

    collection: anything with chroma's `.get(include=[...], where={"is_region": {"$eq": True}})`
    (region_compare.py:51-54).  `score`:
      * "cosine" (default) -- score = cosine similarity, regions scoring below `threshold` are
        dropped: what :266-270 means to do;
      * "reference_distance" -- score = the store's cosine DISTANCE, dropped when below
        `threshold`: what :266-270 literally executes.
    `rows=(row0, nrows)` restricts the source regions (one shard per GPU); candidates are always
    all regions.  Returns a list of dicts, one per source region that has the metadata the
    reference requires (:150-152), in collection order:
      {"id", "parent_image", "type", "area_percentage",
       "similar_regions": [{"id", "score", "weighted_score", "parent_image", "type"}, ...]}   (:340-346)
    """
    if score not in ("cosine", "reference_distance"):
        raise ValueError("score must be 'cosine' or 'reference_distance'")
    all_entries = collection.get(include=["metadatas", "embeddings", "documents"], where={"is_region": {"$eq": True}})
    if not all_entries or not all_entries.get("ids"):
        logger.warning("No regions found in the database. Make sure regions have been processed first.")  # :56-58
        return []
    ids, metas = all_entries["ids"], all_entries["metadatas"]
    n = len(ids)
    engine = engine or default_engine()
    emb = to_unit_bf16(all_entries["embeddings"], engine)

    def parent_of(m):
        return (m or {}).get("parent_image") or (m or {}).get("parent_image_name") or ""

    # group id = parent page; a row without one never equals another row's parent (:257-261)
    gid, group = {}, np.empty(n, dtype=np.int32)
    for r, m in enumerate(metas):
        p = parent_of(m)
        group[r] = gid.setdefault(p, len(gid)) if p else -(r + 1)
    row0, nrows = (0, n) if rows is None else rows
    window = {"min_sim": float(threshold)} if score == "cosine" else {"max_sim": float(1.0 - threshold)}
    idx, sim = engine.neighbours(emb, group, row0=row0, nrows=nrows, fetch=min(top_n * 3, 100), top_n=top_n, **window)
    idx, sim = idx.cpu().numpy(), sim.cpu().numpy().astype(np.float64)

    out = []
    for o in range(nrows):
        r = row0 + o
        meta = metas[r]
        if not meta:
            logger.warning(f"Missing metadata or embedding for region {ids[r]}")  # :123-125
            continue
        parent, rtype = parent_of(meta), meta.get("region_type")
        if not parent or not rtype or not _box_of(meta):
            logger.warning(f"Missing essential metadata for region {ids[r]}")  # :150-152
            continue
        area = meta.get("area_percentage", 0)
        similar = []
        for c, s in zip(idx[o], sim[o]):
            if c < 0:
                break
            cm = metas[c] or {}
            sc = float(s) if score == "cosine" else float(1.0 - s)
            weighted = sc * (area / 100) * (cm.get("area_percentage", 0) / 100) if weight_by_area else sc  # :273-280
            similar.append({"id": ids[c], "score": sc, "weighted_score": weighted,
                            "parent_image": os.path.basename(parent_of(cm)), "type": cm.get("region_type", "unknown")})
        out.append({"id": ids[r], "parent_image": os.path.basename(parent), "type": rtype, "area_percentage": area,
                    "similar_regions": similar})
    return out


def create_region_cross_comparison(collection, top_n=config.REGION_COMPARE_TOP_N, output_path=None, **kwargs):
    """Same entry point as region_compare.py:25; writes one JSON document instead of HTML pages."""
    result = region_neighbours(collection, top_n, **kwargs)
    if output_path:
        os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
        with open(output_path, "w") as fh:
            json.dump({"top_n": top_n, "regions": result}, fh, indent=1)
    return result
