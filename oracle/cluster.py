"""ORACLE (test infrastructure only) -- page clustering, `cluster_images`.

CPU restatement (numpy f64 + stdlib heapq) of
deprecated_package/weighted_region_clustering.py:452-574.  The arithmetic the
reference delegates to third parties (requirements.txt:11 scikit-learn, un-pinned;
scipy transitive) is restated from their published algorithms, for the code path
the bundled golden labels pin:

  scikit-learn >= 1.4: `AgglomerativeClustering(affinity=...)` raises TypeError
  (wrc:499-503) -> fallback `AgglomerativeClustering(n_clusters=k,
  linkage='average').fit(D)` (wrc:504-509) = euclidean metric over the ROWS of D:
    sklearn/cluster/_agglomerative.py:584  scipy.cluster.hierarchy.linkage(D,'average','euclidean')
      scipy/spatial/distance pdist 'euclidean': sequential sum of squares, sqrt
      scipy/cluster/_hierarchy.pyx nn_chain (+ stable sort by height + `label`)
    sklearn/cluster/_agglomerative.py:729-776 `_hc_cut` (heap of node ids)
  sklearn/metrics/cluster/_unsupervised.py silhouette_score(metric='precomputed')

mode="precomputed" restates the path the reference's first `try:` intended
(average linkage directly on D as a distance matrix).

Pinning: KAT = the 19x19 matrix and labels printed in the reference's bundled
report (deprecated_package/output/weighted_clustering/html_report/index.html:603,
:66-286), committed as tests/golden/report_matrix.json, plus outputs of the real
`cluster_images` and of scipy/sklearn on seeded matrices (make_golden.py).
"""
from __future__ import annotations

import heapq

import numpy as np


def pdist_rows_euclidean(X: np.ndarray) -> np.ndarray:
    """Condensed euclidean distances between rows; sum accumulated in column order."""
    X = np.asarray(X, dtype=np.float64)
    n, m = X.shape
    s = np.zeros((n, n), dtype=np.float64)
    for k in range(m):
        d = X[:, None, k] - X[None, :, k]
        s += d * d
    full = np.sqrt(s)
    iu = np.triu_indices(n, 1)
    return full[iu]


def squareform_to_condensed(D: np.ndarray) -> np.ndarray:
    n = D.shape[0]
    return np.asarray(D, dtype=np.float64)[np.triu_indices(n, 1)]


def _cidx(n: int, i: int, j: int) -> int:
    if i > j:
        i, j = j, i
    return n * i - (i * (i + 1)) // 2 + (j - i - 1)


def linkage_average(y: np.ndarray, n: int) -> np.ndarray:
    """scipy `linkage(y, 'average')`: nearest-neighbour chain, then sort + relabel."""
    D = np.array(y, dtype=np.float64)
    size = np.ones(n, dtype=np.int64)
    Z = np.zeros((n - 1, 4), dtype=np.float64)
    chain = np.zeros(n, dtype=np.int64)
    clen = 0
    x = y_ = 0
    for k in range(n - 1):
        if clen == 0:
            clen = 1
            for i in range(n):
                if size[i] > 0:
                    chain[0] = i
                    break
        while True:
            x = int(chain[clen - 1])
            if clen > 1:
                y_ = int(chain[clen - 2])
                cur = D[_cidx(n, x, y_)]
            else:
                cur = np.inf
            for i in range(n):
                if size[i] == 0 or x == i:
                    continue
                dist = D[_cidx(n, x, i)]
                if dist < cur:
                    cur = dist
                    y_ = i
            if clen > 1 and y_ == chain[clen - 2]:
                break
            chain[clen] = y_
            clen += 1
        clen -= 2
        if x > y_:
            x, y_ = y_, x
        nx, ny = int(size[x]), int(size[y_])
        Z[k] = (x, y_, cur, nx + ny)
        size[x] = 0
        size[y_] = nx + ny
        for i in range(n):
            ni = size[i]
            if ni == 0 or i == y_:
                continue
            D[_cidx(n, i, y_)] = (nx * D[_cidx(n, i, x)] + ny * D[_cidx(n, i, y_)]) / (nx + ny)
    order = np.argsort(Z[:, 2], kind="stable")
    Z = Z[order]
    # `label`: union-find relabelling to scipy's node numbering
    parent = np.arange(2 * n - 1)
    usize = np.ones(2 * n - 1, dtype=np.int64)
    nxt = n

    def find(a):
        p = a
        while parent[a] != a:
            a = parent[a]
        while parent[p] != a:
            p, parent[p] = parent[p], a
        return a

    for i in range(n - 1):
        a, b = int(Z[i, 0]), int(Z[i, 1])
        ra, rb = find(a), find(b)
        Z[i, 0], Z[i, 1] = (ra, rb) if ra < rb else (rb, ra)
        parent[ra] = nxt
        parent[rb] = nxt
        usize[nxt] = usize[ra] + usize[rb]
        Z[i, 3] = usize[nxt]
        nxt += 1
    return Z


def _descendants(node: int, children: np.ndarray, n_leaves: int):
    out, stack = [], [node]
    while stack:
        v = stack.pop()
        if v < n_leaves:
            out.append(v)
        else:
            stack.extend(children[v - n_leaves])
    return out


def hc_cut(n_clusters: int, children: np.ndarray, n_leaves: int) -> np.ndarray:
    """sklearn `_hc_cut`: label i = position in the heap array after k-1 splits."""
    nodes = [-(int(max(children[-1])) + 1)]
    for _ in range(n_clusters - 1):
        these = children[-nodes[0] - n_leaves]
        heapq.heappush(nodes, -int(these[0]))
        heapq.heappushpop(nodes, -int(these[1]))
    label = np.zeros(n_leaves, dtype=np.int64)
    for i, node in enumerate(nodes):
        label[_descendants(-node, children, n_leaves)] = i
    return label


def agglomerative_labels(D: np.ndarray, k: int, mode: str = "reference_fallback") -> np.ndarray:
    n = D.shape[0]
    y = pdist_rows_euclidean(D) if mode == "reference_fallback" else squareform_to_condensed(D)
    Z = linkage_average(y, n)
    children = Z[:, :2].astype(np.int64)
    return hc_cut(k, children, n)


def silhouette_precomputed(D: np.ndarray, labels: np.ndarray) -> float:
    """sklearn silhouette_score(D, labels, metric='precomputed') (mean over samples)."""
    n = D.shape[0]
    k = int(labels.max()) + 1
    freq = np.bincount(labels, minlength=k).astype(np.float64)
    if not (1 < len(np.unique(labels)) < n):
        raise ValueError("Number of labels is invalid")
    clust = np.zeros((n, k), dtype=np.float64)
    for i in range(n):
        clust[i] = np.bincount(labels, weights=D[i], minlength=k)
    rows = np.arange(n)
    intra = clust[rows, labels].copy()
    clust[rows, labels] = np.inf
    clust /= freq
    inter = clust.min(axis=1)
    denom = (freq - 1).take(labels, mode="clip")
    with np.errstate(divide="ignore", invalid="ignore"):
        intra /= denom
        sil = inter - intra
        sil /= np.maximum(intra, inter)
    return float(np.mean(np.nan_to_num(sil)))


def cluster_images(similarity_matrix: np.ndarray, image_names, n_clusters=None, mode: str = "reference_fallback"):
    """wrc:452-574 restated (the caller's matrix IS mutated, as the reference does, :457)."""
    S = similarity_matrix
    np.fill_diagonal(S, 1.0)
    D = 1.0 - S
    if not isinstance(D, np.ndarray) or D.size == 0 or D.shape[0] != D.shape[1] or np.any(np.isnan(D)):
        return None
    P = len(image_names)
    if n_clusters is None:
        nonzero_pairs = int(np.sum(S > 0.01)) - S.shape[0]  # :482
        max_clusters = min(3, P) if nonzero_pairs < 10 else min(10, P)
        best_score, best_k = -1, 2
        for k in range(2, max_clusters + 1):
            labels = agglomerative_labels(D, k, mode)
            if len(np.unique(labels)) > 1:
                try:
                    score = silhouette_precomputed(D, labels)
                except ValueError:
                    continue  # wrc:524-526
                if score > best_score:
                    best_score, best_k = score, k
        n_clusters = best_k
    labels = agglomerative_labels(D, n_clusters, mode)
    clusters: dict[int, list] = {}
    for i, lab in enumerate(labels):
        clusters.setdefault(int(lab), []).append(image_names[i])
    cohesion = {}
    for lab, imgs in clusters.items():
        idx = [image_names.index(im) for im in imgs]
        if len(idx) > 1:
            sub = S[np.ix_(idx, idx)]
            mask = ~np.eye(len(idx), dtype=bool)
            cohesion[lab] = float(np.mean(sub[mask]))
        else:
            cohesion[lab] = 0.0
    return {
        "n_clusters": n_clusters,
        "clusters": {str(k): v for k, v in clusters.items()},
        "cluster_cohesion": cohesion,
        "labels": [int(v) for v in labels],
    }
