"""ORACLE (test infrastructure only) -- cosine compare + area-weighted page matrix.

CPU restatement (numpy f64, literal Python loops -- small cases only) of
  * the vector-store query every reference "compare" goes through:
    deprecated_package/weighted_region_clustering.py:73-95 (`safe_query`),
    region_compare.py:165-170, cross_compare.py:119-123 -- ChromaDB
    `collection.query`, an approximate HNSW kNN; restated as the EXACT
    brute-force ranking those calls sample from;
  * `compute_image_similarity_matrix`, weighted_region_clustering.py:97-254;
  * the neighbour-report ranking rules of region_compare.py:160-353 and
    cross_compare.py:109-235 (self / same-parent / same-prefix exclusion).

Third-party dependency that is absent from /root/reference and from this image:
chromadb (requirements.txt:5, un-pinned) -> hnswlib.  Its metric in effect is
unverifiable offline (keys spelled `hnsw_space`), so
both candidates are restated: metric="cosine" (d = 1 - cos, the stated intent,
db_operations.py:29) and metric="sqeuclidean" (d = 2 - 2cos on unit vectors,
chroma's default space).  "parity unpinned" at that boundary; pinned instead on
the reference's own call sites by running the real
`compute_image_similarity_matrix` over a duck-typed brute-force collection
(tests/golden/make_golden.py) and committing inputs + outputs.
"""
from __future__ import annotations

import numpy as np

REGION_TYPES_TO_PROCESS = (  # deprecated_package/config.py:67-74
    "title",
    "plain_text",
    "figure",
    "figure_caption",
    "table",
    "table_caption",
)


def cosine_matrix(a: np.ndarray, b: np.ndarray | None = None) -> np.ndarray:
    """All-pairs cosine in f64: rows are normalised here, so any scale is accepted."""
    a = np.asarray(a, dtype=np.float64)
    b = a if b is None else np.asarray(b, dtype=np.float64)
    an = a / np.linalg.norm(a, axis=1, keepdims=True)
    bn = b / np.linalg.norm(b, axis=1, keepdims=True)
    return an @ bn.T


def distances(q: np.ndarray, e: np.ndarray, metric: str = "cosine") -> np.ndarray:
    """Distance of one query row to every row of e, as chroma would report it."""
    cos = cosine_matrix(q[None, :], e)[0]
    if metric == "cosine":
        return 1.0 - cos
    if metric == "sqeuclidean":
        return 2.0 - 2.0 * cos
    raise ValueError(metric)


def same_prefix_mask(names, prefix_length: int = 20) -> np.ndarray:
    """weighted_region_clustering.py:179-186: first `prefix_length` characters equal."""
    pre = [n[: min(prefix_length, len(n))] for n in names]
    P = len(names)
    m = np.zeros((P, P), dtype=bool)
    for i in range(P):
        for j in range(P):
            m[i, j] = pre[i] == pre[j]
    return m


def pair_terms(emb, area_percentage, regions_i, rows_j, n_valid_j, *, metric="cosine", effective_threshold=0.1,
               max_query_regions=10, top_k=10, sim=None):
    """Weighted terms of ONE page pair (wrc:199-226): the list whose np.sum is S[i,j] before normalisation."""
    weighted = []
    n_results = min(top_k, n_valid_j)  # :210
    for r in regions_i[:max_query_regions]:  # :199
        area_i = area_percentage[r] / 100.0
        if area_i == 0:
            continue
        if sim is None:
            d = distances(emb[r], emb[rows_j], metric)
        else:
            c64 = np.asarray(sim[r, rows_j], dtype=np.float64)
            d = 1.0 - c64 if metric == "cosine" else 2.0 - 2.0 * c64
        order = np.argsort(d, kind="stable")[:n_results]
        for k in order:
            dist = float(d[k])
            area_j = area_percentage[rows_j[k]] / 100.0
            if dist <= (1.0 - effective_threshold) and area_j > 0:  # :223
                weighted.append((1.0 - dist) * area_i * area_j)
    return weighted


def compute_image_similarity_matrix(
    emb: np.ndarray,
    area_percentage: np.ndarray,
    page_of: np.ndarray,
    names,
    region_types=None,
    *,
    metric: str = "cosine",
    effective_threshold: float = 0.1,
    skip_same_prefix: bool = True,
    prefix_length: int = 20,
    max_query_regions: int = 10,
    top_k: int = 10,
    normalise: bool = True,
    sim: np.ndarray | None = None,
):
    """weighted_region_clustering.py:97-254 restated over an in-memory region table.

    `sim` (optional, [N,N]) supplies the cosine values instead of recomputing them from
    `emb`: the parity tests pass the HIP cosine kernel's own f32 output here, so that the
    top-k / threshold / reduction logic is compared decision for decision (a 1e-7
    difference between an f32 and an f64 dot product can legitimately reorder a near-tie).

    emb[N,D], area_percentage[N] (0-100, region_processor.py:89-93), page_of[N]
    (index into `names`, rows in collection.get order), names[P] (page basenames
    in the order of `image_paths`).  Returns (S[P,P] f64, names) or (None, None)
    when the table is empty (:116-118).
    """
    N = len(page_of)
    if N == 0:
        return None, None
    P = len(names)
    S = np.zeros((P, P), dtype=np.float64)
    area_percentage = np.asarray(area_percentage, dtype=np.float64)
    area = area_percentage / 100.0  # :138
    ok_type = np.ones(N, dtype=bool) if region_types is None else np.array([t in REGION_TYPES_TO_PROCESS for t in region_types])
    valid = (area_percentage > 0) & ok_type  # :136
    rows_of = [np.flatnonzero(page_of == p) for p in range(P)]  # every DB row of the page (query `where`)
    regions_of = [r[valid[r]] for r in rows_of]  # image_to_regions (:137)
    for i in range(P):
        ri = regions_of[i]
        if len(ri) == 0:
            continue
        for j in range(i + 1, P):
            rj = regions_of[j]
            if len(rj) == 0:
                continue
            if skip_same_prefix:
                if names[i][: min(prefix_length, len(names[i]))] == names[j][: min(prefix_length, len(names[j]))]:
                    continue
            weighted = pair_terms(emb, area_percentage, ri, rows_of[j], len(rj), metric=metric, effective_threshold=effective_threshold,
                                  max_query_regions=max_query_regions, top_k=top_k, sim=sim)
            if weighted:
                s = np.sum(weighted)
                S[i, j] = s
                S[j, i] = s
    if normalise:
        mx = np.max(S - np.diag(np.diag(S)))  # :246
        if mx > 0:
            off = ~np.eye(P, dtype=bool)
            S[off] = S[off] / mx
        np.fill_diagonal(S, 1.0)  # :252
    return S, list(names)


def neighbour_lists(
    emb: np.ndarray | None,
    group_of: np.ndarray | None,
    *,
    top_n: int = 10,
    fetch: int | None = None,
    area_percentage: np.ndarray | None = None,
    metric: str = "cosine",
    sim: np.ndarray | None = None,
    rows: range | None = None,
    keep_self: bool = False,
    min_sim: float = -np.inf,
    max_sim: float = np.inf,
):
    """Selection loop of region_compare.py:160-353.

    Per query row r: take the `fetch` = min(3*top_n, 100) nearest rows (:163; r itself is among
    them, as the store's query returns it) in stable ascending-distance order, drop r (:244), drop
    rows of r's group = same parent page (:260), drop similarities outside [min_sim, max_sim]
    (the score window of :269 -- G2: the reference applies it to the raw distance, i.e.
    max_sim = 1 - 0.3), keep the first `top_n` (:352).  `sim` overrides the f64 cosine matrix (tests
    pass the kernel's own f32 values so that ties break on identical numbers).
    Returns (idx[R, top_n] int64 padded with -1, sim[R, top_n] f64 cosine, weighted[R, top_n] =
    sim*a_src/100*a_tgt/100 (:273-278)).
    """
    C = cosine_matrix(emb) if sim is None else np.asarray(sim)
    N = C.shape[1]
    rows = range(N) if rows is None else rows
    fetch = min(3 * top_n, 100) if fetch is None else fetch
    idx = np.full((len(rows), top_n), -1, dtype=np.int64)
    out = np.zeros((len(rows), top_n), dtype=np.float64)
    wsim = np.zeros((len(rows), top_n), dtype=np.float64)
    for o, r in enumerate(rows):
        row = C[o] if C.shape[0] == len(rows) and C.shape[0] != N else C[r]
        d = 1.0 - row if metric == "cosine" else 2.0 - 2.0 * row
        order = np.argsort(d, kind="stable")[:fetch]
        k = 0
        for c in order:
            if c == r and not keep_self:
                continue
            if group_of is not None and c != r and group_of[c] == group_of[r]:
                continue
            if not (min_sim <= row[c] <= max_sim):
                continue
            idx[o, k] = c
            out[o, k] = row[c]
            if area_percentage is not None:
                wsim[o, k] = row[c] * (area_percentage[r] / 100.0) * (area_percentage[c] / 100.0)
            k += 1
            if k == top_n:
                break
    return idx, out, wsim


def image_neighbour_lists(sim: np.ndarray, filenames, *, top_n: int = 5, fetch: int | None = None):
    """Selection loop of cross_compare.py:109-235: `fetch` = min(5*top_n, 100) nearest images (:117),
    drop the source (:176) and every image whose filename starts with the SOURCE's prefix of
    max(1, int(len*0.2)) characters (:109-110, :200-206), keep the first top_n (:234).
    Returns a list (one per image) of [(index, similarity), ...]."""
    sim = np.asarray(sim)
    N = sim.shape[0]
    fetch = min(5 * top_n, 100) if fetch is None else fetch
    out = []
    for r in range(N):
        plen = max(1, int(len(filenames[r]) * 0.2))
        src_prefix = filenames[r][:plen]
        order = np.argsort(1.0 - sim[r], kind="stable")[:fetch]
        lst = []
        for c in order:
            if c == r:
                continue
            cand = filenames[c]
            if (cand[:plen] if len(cand) >= plen else cand) == src_prefix:
                continue
            lst.append((int(c), float(sim[r, c])))
            if len(lst) == top_n:
                break
        out.append(lst)
    return out
