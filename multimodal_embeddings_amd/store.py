"""Persistence in the shapes the reference's stages exchange.

* region rows exactly as `collection.upsert(ids, embeddings, documents, metadatas)` receives them
  (region_processor.py:141-149) -> one `.npz` (vectors) + one `.json` (ids, documents, metadatas);
* the artefacts `weighted_region_clustering.main` leaves behind
  (deprecated_package/weighted_region_clustering.py:871-892): `similarity_matrix.npy`,
  `image_names.json`, `clustering_results.json`;
* resume bookkeeping keyed like progress_tracker.py (`completed_*` lists of ids), but append-only:
  O(1) per finished item instead of rewriting the whole JSON list each time.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .weighted_region_clustering import RegionCollection, cluster_images, compute_image_similarity_matrix


def save_collection(collection, path_stem):
    """Write `<stem>.npz` (embeddings float32[N, D]) and `<stem>.json` (ids, documents, metadatas)."""
    rows = collection.get(include=["metadatas", "embeddings", "documents"])
    emb = np.asarray(rows["embeddings"], dtype=np.float32).reshape(len(rows["ids"]), -1)
    os.makedirs(os.path.dirname(os.path.abspath(path_stem)), exist_ok=True)
    np.savez_compressed(path_stem + ".npz", embeddings=emb)
    with open(path_stem + ".json", "w") as fh:
        json.dump({"ids": rows["ids"], "documents": rows.get("documents") or [None] * len(rows["ids"]), "metadatas": rows["metadatas"]}, fh)
    return len(rows["ids"])


def load_collection(path_stem) -> RegionCollection:
    """Inverse of `save_collection`; rows come back in their original order (the 'first 10 regions'
    rule of wrc:199 and the tie order of equal distances depend on it)."""
    with open(path_stem + ".json") as fh:
        meta = json.load(fh)
    emb = np.load(path_stem + ".npz")["embeddings"]
    col = RegionCollection()
    col.upsert(ids=meta["ids"], embeddings=[e for e in emb], documents=meta["documents"], metadatas=meta["metadatas"])
    return col


def save_clustering_outputs(output_dir, similarity_matrix, image_names, clustering_results):
    """The three files of wrc:871-892, same names and encodings (np.save; json.dump; json.dump indent=2,
    which turns the int keys of `cluster_cohesion` into strings exactly as the reference's file has them)."""
    os.makedirs(output_dir, exist_ok=True)
    np.save(os.path.join(output_dir, "similarity_matrix.npy"), similarity_matrix)
    with open(os.path.join(output_dir, "image_names.json"), "w") as fh:
        json.dump(image_names, fh)
    if clustering_results is not None:
        with open(os.path.join(output_dir, "clustering_results.json"), "w") as fh:
            json.dump(clustering_results, fh, indent=2)


def run_weighted_clustering(collection, image_paths, output_dir, n_clusters=None, *, skip_same_prefix=True, prefix_length=20,
                            engine=None):
    """Body of `weighted_region_clustering.main` (wrc:857-892) without plots / HTML: page matrix ->
    files -> clusters -> file.  Returns (S, names, results) or None where the reference logs and exits."""
    S, names = compute_image_similarity_matrix(collection, image_paths, skip_same_prefix=skip_same_prefix,
                                               prefix_length=prefix_length, engine=engine)
    if S is None or S.size == 0 or not names:
        return None  # wrc:866-868
    save_clustering_outputs(output_dir, S, names, None)
    results = cluster_images(S, names, n_clusters=n_clusters, engine=engine)  # mutates S's diagonal like wrc:457
    if results is None:
        return None  # wrc:882-884
    save_clustering_outputs(output_dir, S, names, results)
    return S, names, results


class ProgressLog:
    """Idempotent resume: `done(key)` / `mark(key)` over an append-only JSON-lines file.

    progress_tracker.py keeps `{"completed_...": [ids]}` and rewrites the whole file for every
    finished item (O(n^2) bytes over a run); here finishing an item appends one line.
    `import_reference(path, field)` seeds the log from one of the reference's progress files."""

    def __init__(self, path):
        self.path = path
        self._done = set()
        if os.path.exists(path):
            with open(path) as fh:
                for line in fh:
                    line = line.strip()
                    if line:
                        self._done.add(json.loads(line))
        else:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def done(self, key) -> bool:
        return key in self._done

    def mark(self, key):
        if key in self._done:
            return
        self._done.add(key)
        with open(self.path, "a") as fh:
            fh.write(json.dumps(key) + "\n")

    def import_reference(self, path, field):
        with open(path) as fh:
            for key in json.load(fh).get(field, []):
                self.mark(key)

    def __len__(self):
        return len(self._done)
