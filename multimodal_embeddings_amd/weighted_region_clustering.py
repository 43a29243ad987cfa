"""Host mirror of deprecated_package/weighted_region_clustering.py:97-254 and :452-574.

    S, names = compute_image_similarity_matrix(collection, image_paths)
    result   = cluster_images(S, names)

Same names, argument meaning and return shapes as the reference; the page-pair loop
runs as kernel K10 (`mme_page_similarity`) over exact brute-force cosine instead of
P^2*10 ChromaDB queries, and the agglomerative clustering runs as kernel K11
(`mme_cluster_pages`).  Behaviours deliberately kept:
  G5  `similarity_threshold` is accepted and ignored; the effective threshold is 0.1
      (wrc:151) -- exposed as the real parameter `effective_threshold`;
  G6  only the first 10 valid regions of the lower-index page query (wrc:199);
  G3  `cluster_images` defaults to mode="reference_fallback" (euclidean over the rows
      of D, what scikit-learn >= 1.4 executes through wrc:504-509, and what the bundled
      golden labels pin); mode="precomputed" is the path the first `try:` intended;
  G11 `cluster_images` mutates the caller's matrix diagonal in place (wrc:457).
Behaviours deliberately NOT kept: the per-pair JSON progress file and its resume bug
(G7, wrc:189-192): pairs are always recomputed.
"""
from __future__ import annotations

import logging
import os
from collections import defaultdict

import numpy as np

from . import config
from .cross_compare import default_engine

logger = logging.getLogger("multimodal_embeddings_amd")


def _where_mask(metadatas, where):
    """chroma `where` filter -> list[bool].  Supported (all the reference's call sites use the first form):
    {"key": {"$eq": v}}, {"key": {"$ne": v}}, {"key": v}, {"$and": [...]}, {"$or": [...]}."""
    n = len(metadatas)
    if not where:
        return [True] * n
    if len(where) != 1:
        raise ValueError("where: one condition per level; combine conditions with $and / $or")
    (key, cond), = where.items()
    if key in ("$and", "$or"):
        parts = [_where_mask(metadatas, w) for w in cond]
        return [all(p[i] for p in parts) if key == "$and" else any(p[i] for p in parts) for i in range(n)]
    if isinstance(cond, dict):
        (op, val), = cond.items()
    else:
        op, val = "$eq", cond
    if op == "$eq":
        return [m is not None and key in m and m[key] == val for m in metadatas]
    if op == "$ne":
        return [m is not None and m.get(key) != val for m in metadatas]
    raise ValueError(f"where operator {op!r} is not supported")


class RegionCollection:
    """In-memory stand-in for the chroma collection (db_operations.py:17-63) with the calls the hot path makes.

    Holds what region_processor.py:141-149 upserts: ids, embeddings, metadatas (with
    `parent_image_name`, `region_type`, `area_percentage`, `is_region`).  `upsert` / `add`, `get`, `count`
    and `query` have chroma's argument names and return shapes, so the reference's call sites
    (wrc:79-84, region_compare.py:165-170, cross_compare.py:119-123, demo_queries.py:61-66) run unchanged
    against it.  `query` is exact (no HNSW): the ranking is computed on the GPU by kernel K12
    (`mme_neighbours`) over the L2-normalised bf16 rows -- distance ascending, ties in insertion order.
    `metric`: "cosine" (d = 1 - cos; the reference's stated intent, db_operations.py:29) or "sqeuclidean"
    (d = 2 - 2 cos on unit vectors, chroma's default space).
    """

    def __init__(self, metric="cosine", engine=None):
        if metric not in ("cosine", "sqeuclidean"):
            raise ValueError("metric must be 'cosine' or 'sqeuclidean'")
        self.metric = metric
        self.engine = engine
        self.metadata = {"hnsw_space": metric}
        self.ids, self.embeddings, self.metadatas, self.documents = [], [], [], []
        self._pos = {}
        self._device_rows = None  # unit bf16 rows of all embeddings, rebuilt after an upsert

    def upsert(self, ids, embeddings, documents=None, metadatas=None):
        documents = documents or [None] * len(ids)
        metadatas = metadatas or [None] * len(ids)
        for i, e, d, m in zip(ids, embeddings, documents, metadatas):
            if i in self._pos:
                k = self._pos[i]
                self.embeddings[k], self.documents[k], self.metadatas[k] = e, d, m
            else:
                self._pos[i] = len(self.ids)
                self.ids.append(i)
                self.embeddings.append(e)
                self.documents.append(d)
                self.metadatas.append(m)
        self._device_rows = None

    add = upsert

    def update(self, ids, embeddings=None, documents=None, metadatas=None):
        """chroma's `Collection.update` (image_processor.py:219-224): change the given fields of rows that EXIST; an unknown
        id is skipped with a warning (chroma logs and ignores it), a field passed as None keeps its value."""
        for k, i in enumerate(ids):
            if i not in self._pos:
                logger.warning(f"update: id {i!r} not in the collection")
                continue
            r = self._pos[i]
            if embeddings is not None:
                self.embeddings[r] = embeddings[k]
            if documents is not None:
                self.documents[r] = documents[k]
            if metadatas is not None:
                self.metadatas[r] = metadatas[k]
        if embeddings is not None:
            self._device_rows = None

    def modify(self, name=None, metadata=None):
        """chroma's `Collection.modify` as db_operations.py:50-52 uses it: replace the collection-level metadata."""
        if metadata is not None:
            self.metadata = dict(metadata)
        if name is not None:
            self.name = name

    def count(self):
        return len(self.ids)

    def get(self, ids=None, include=None, where=None):
        rows = range(len(self.ids)) if ids is None else [self._pos[i] for i in ids if i in self._pos]
        if where:
            keep = _where_mask(self.metadatas, where)
            rows = [r for r in rows if keep[r]]
        rows = list(rows)
        return {
            "ids": [self.ids[r] for r in rows],
            "embeddings": [self.embeddings[r] for r in rows],
            "metadatas": [self.metadatas[r] for r in rows],
            "documents": [self.documents[r] for r in rows],
        }

    def query(self, query_embeddings=None, n_results=10, where=None, include=("metadatas", "documents", "distances"), engine=None):
        """chroma's `Collection.query`: for every query vector the `n_results` nearest stored vectors that pass
        `where`, as lists of lists (one inner list per query), distance ascending.

        Exact brute force on the GPU: the candidates that pass `where` and the query vectors are stacked into one
        table of unit bf16 rows; kernel K12 ranks the query rows against it with the queries' own group masked out
        (so a query never returns itself or another query).  Up to 128 results per query come from K12's streaming
        selection (k + number of queries <= 128); larger requests sort a K9 cosine block."""
        from .cross_compare import to_unit_bf16

        if query_embeddings is None or len(query_embeddings) == 0:
            raise ValueError("query_embeddings is required (text queries need the language tower, which is out of scope)")
        keep = _where_mask(self.metadatas, where)
        rows = [r for r in range(len(self.ids)) if keep[r] and self.embeddings[r] is not None and len(self.embeddings[r]) > 0]
        nq, n_c = len(query_embeddings), len(rows)
        k = max(0, min(int(n_results), n_c))
        out = {"ids": [[] for _ in range(nq)], "distances": None, "metadatas": None, "documents": None, "embeddings": None}
        for key in ("distances", "metadatas", "documents", "embeddings"):
            if key in include:
                out[key] = [[] for _ in range(nq)]
        if k == 0:
            return out
        engine = engine or self.engine or default_engine()
        t = engine.torch
        if self._device_rows is None:
            have = [r for r in range(len(self.ids)) if self.embeddings[r] is not None and len(self.embeddings[r]) > 0]
            self._device_rows = (have, to_unit_bf16([self.embeddings[r] for r in have], engine))
        have, table = self._device_rows
        if len(rows) != len(have):
            where_of = {r: i for i, r in enumerate(have)}
            sel = t.tensor([where_of[r] for r in rows], dtype=t.long, device=table.device)
            cand = table.index_select(0, sel)
        else:
            cand = table
        q = to_unit_bf16(query_embeddings, engine)
        if k + nq <= 128:
            # K12 keeps the best `fetch` rows BEFORE it drops the query's own group: the nq query rows of the stacked
            # table (the query itself at cosine 1, possibly the other queries) may all rank inside, so fetch k + nq
            stacked = t.cat([cand, q], dim=0)
            group = t.zeros(n_c + nq, dtype=t.int32, device=stacked.device)
            group[n_c:] = 1
            idx, sim = engine.neighbours(stacked, group, row0=n_c, nrows=nq, fetch=k + nq, top_n=k)
        else:
            block = engine.cosine(q, cand)
            order = t.sort(block, dim=1, descending=True, stable=True)
            idx, sim = order.indices[:, :k], order.values[:, :k]
        idx, sim = idx.cpu().numpy(), sim.cpu().numpy().astype(np.float64)
        dist = 1.0 - sim if self.metric == "cosine" else 2.0 - 2.0 * sim
        for qi in range(nq):
            picks = [rows[c] for c in idx[qi] if c >= 0]
            out["ids"][qi] = [self.ids[r] for r in picks]
            if out["distances"] is not None:
                out["distances"][qi] = [float(d) for d in dist[qi][: len(picks)]]
            if out["metadatas"] is not None:
                out["metadatas"][qi] = [self.metadatas[r] for r in picks]
            if out["documents"] is not None:
                out["documents"][qi] = [self.documents[r] for r in picks]
            if out["embeddings"] is not None:
                out["embeddings"][qi] = [self.embeddings[r] for r in picks]
        return out


def safe_query(collection, query_embedding, n_results, where_clause, max_retries=3):
    """wrc:73-95: the retry ladder guards an hnswlib failure mode ("Cannot return the results in a contigious 2D
    array") that an exact ranking does not have; kept so call sites port unchanged."""
    for attempt in range(max_retries):
        try:
            return collection.query(query_embeddings=[query_embedding], n_results=n_results,
                                    include=["metadatas", "documents", "distances"], where=where_clause)
        except RuntimeError as e:
            if "Cannot return the results in a contigious 2D array" in str(e) and attempt < max_retries - 1:
                n_results = max(1, int(n_results * 0.8))
                continue
            raise
    return None


def same_prefix_skip(image_names, prefix_length=config.PREFIX_LENGTH):
    """uint8[P,P]: 1 where both names share their first `prefix_length` characters (wrc:179-186)."""
    pre = np.array([n[: min(prefix_length, len(n))] for n in image_names], dtype=object)
    return (pre[:, None] == pre[None, :]).astype(np.uint8)


def build_page_table(all_entries, image_names):
    """Group the collection's rows by parent page in `image_names` order (wrc:120-139).

    Returns (emb float32 [N,D], area_percentage f64 [N], valid uint8 [N], page_offs int32 [P+1]);
    row order inside a page is collection order, which the "first 10 regions" rule (G6)
    and the tie order of equal distances depend on.
    """
    name_to_idx = {n: i for i, n in enumerate(image_names)}
    per_page = defaultdict(list)
    for r, meta in enumerate(all_entries["metadatas"]):
        emb = all_entries["embeddings"][r]
        if not meta or emb is None or len(emb) == 0:
            continue
        p = name_to_idx.get(meta.get("parent_image_name"))
        if p is None:
            continue
        per_page[p].append(r)
    rows, offs = [], [0]
    for p in range(len(image_names)):
        rows.extend(per_page.get(p, []))
        offs.append(len(rows))
    metas = [all_entries["metadatas"][r] for r in rows]
    area = np.array([float(m.get("area_percentage", 0) or 0) for m in metas], dtype=np.float64)
    ok_type = np.array([m.get("region_type") in config.REGION_TYPES_TO_PROCESS for m in metas], dtype=bool)
    valid = ((area > 0) & ok_type).astype(np.uint8)
    emb = np.asarray([all_entries["embeddings"][r] for r in rows], dtype=np.float32)
    return emb, area, valid, np.asarray(offs, dtype=np.int32)


def page_similarity_from_table(emb, area_percentage, valid, page_offs, image_names, *, metric="cosine",
                               effective_threshold=config.EFFECTIVE_THRESHOLD, skip_same_prefix=True,
                               prefix_length=config.PREFIX_LENGTH, max_query=config.PAGE_QUERY_REGIONS,
                               top_k=config.PAGE_TOP_K, normalise=True, engine=None, pair_range=None):
    """Device entry: emb may be a bf16 CUDA tensor of unit rows (straight from the embedder).

    pair_range=(lo, hi) computes one rank's share of the page pairs (raw, zero elsewhere): see
    `dist.page_similarity_sharded`."""
    from .cross_compare import to_unit_bf16

    engine = engine or default_engine()
    t = engine.torch
    dev = t.device(f"cuda:{engine.device}")
    e = to_unit_bf16(emb, engine)
    a = t.from_numpy(np.ascontiguousarray(area_percentage, dtype=np.float64)).to(dev)
    v = t.from_numpy(np.ascontiguousarray(valid, dtype=np.uint8)).to(dev)
    skip = None
    if skip_same_prefix:
        skip = t.from_numpy(same_prefix_skip(image_names, prefix_length)).to(dev)
    S = engine.page_similarity(e, a, v, page_offs, skip, max_query=max_query, top_k=top_k,
                               max_dist=1.0 - effective_threshold, metric={"cosine": 0, "sqeuclidean": 1}[metric],
                               normalise=normalise, pair_range=pair_range)
    return S


def compute_image_similarity_matrix(collection, image_paths, similarity_threshold=config.REGION_SIMILARITY_THRESHOLD,
                                    skip_same_prefix=True, prefix_length=config.PREFIX_LENGTH, *, metric="cosine",
                                    effective_threshold=config.EFFECTIVE_THRESHOLD, engine=None):
    """wrc:97-254.  Returns (S float64 [P,P], image_names) or (None, None) without regions."""
    image_names = [os.path.basename(p) for p in image_paths]
    all_entries = collection.get(include=["metadatas", "embeddings"], where={"is_region": {"$eq": True}})
    if not all_entries or len(all_entries["metadatas"]) == 0:
        logger.warning("No regions found in the database. Make sure regions have been processed first.")
        return None, None
    emb, area, valid, offs = build_page_table(all_entries, image_names)
    logger.info(f"Found {int(valid.sum())} regions across {len(image_names)} images")
    logger.info(f"Using similarity threshold: {effective_threshold} (original: {similarity_threshold})")
    if emb.shape[0] == 0:
        S = np.zeros((len(image_names), len(image_names)))
        np.fill_diagonal(S, 1.0)
        return S, image_names
    S = page_similarity_from_table(emb, area, valid, offs, image_names, metric=metric, effective_threshold=effective_threshold,
                                   skip_same_prefix=skip_same_prefix, prefix_length=prefix_length, engine=engine)
    return S.cpu().numpy(), image_names


def cluster_images(similarity_matrix, image_names, n_clusters=None, *, mode="reference_fallback", engine=None):
    """wrc:452-574.  Returns {n_clusters, clusters, cluster_cohesion, labels} or None."""
    try:
        if not isinstance(similarity_matrix, np.ndarray) or similarity_matrix.size == 0:
            logger.error("Distance matrix is not a valid numpy array")
            return None
        if similarity_matrix.ndim != 2 or similarity_matrix.shape[0] != similarity_matrix.shape[1]:
            logger.error("Distance matrix is not square")
            return None
        np.fill_diagonal(similarity_matrix, 1.0)  # wrc:457 (mutates the caller's array, G11)
        if np.any(np.isnan(similarity_matrix)):
            logger.error("Distance matrix contains NaN values")
            return None
        P = len(image_names)
        if similarity_matrix.shape[0] != P:
            raise ValueError("image_names and similarity_matrix disagree")
        engine = engine or default_engine()
        labels, k, scores = engine.cluster_pages(similarity_matrix, n_clusters, mode)
        for kk, sc in scores:
            logger.info(f"Testing {kk} clusters: silhouette score = {sc:.4f}")
        clusters = defaultdict(list)
        for i, lab in enumerate(labels):
            clusters[int(lab)].append(image_names[i])
        # intra-cluster cohesion (wrc:549-561): O(P^2) bookkeeping on the finished labels
        cohesion = {}
        for lab, imgs in clusters.items():
            idx = [image_names.index(im) for im in imgs]
            if len(idx) > 1:
                sub = similarity_matrix[np.ix_(idx, idx)]
                mask = ~np.eye(sub.shape[0], dtype=bool)
                cohesion[lab] = float(np.mean(sub[mask])) if np.any(mask) else 0
            else:
                cohesion[lab] = 0.0
        return {
            "n_clusters": k,
            "clusters": {str(kk): v for kk, v in clusters.items()},
            "cluster_cohesion": cohesion,
            "labels": [int(v) for v in labels],
        }
    except Exception as e:  # wrc:570-574
        logger.error(f"Error during clustering: {str(e)}")
        return None
