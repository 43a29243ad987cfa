#!/usr/bin/env python3
"""Generate the committed golden vectors (run in the BUILD CONTAINER only).

    python tests/golden/make_golden.py

Imports the reference's own Python from /root/reference (recipe: SURVEY.md
Appendix C.3 -- heavy optional imports are replaced by empty module objects; no
reference source is copied) and the third-party libraries the reference
delegates to (transformers ViT / Mllama image processor, Pillow, scikit-learn,
scipy), runs them on seeded inputs or on the data files the reference bundles,
and writes inputs + expected outputs as small fixtures next to this script.
Nothing here runs on the GPU box: /root/reference does not exist there, the
fixtures do.

Fixtures written
  report_matrix.json     19x19 page matrix (3 dp), page names, labels, k -- parsed from the
                         reference's bundled report (a data file): KAT for cluster_images
  region_table.json      the 19 bundled region_cache/*.json files, compacted (data files)
  cluster_cases.npz      seeded S matrices -> labels/k/cohesion from the REAL cluster_images
  linkage_cases.npz      seeded matrices -> scipy linkage Z, sklearn labels, silhouette
  pagesim_cases.npz      region tables + seeded unit vectors -> S from the REAL
                         compute_image_similarity_matrix over a brute-force collection
  pagesim_bf16_cases.npz the same table (and a 48-page D = 768 one) with the rows ROUNDED TO BF16, as the device holds them, and
                         inner products of the stored rows as the collection's cosine -> S from the REAL
                         compute_image_similarity_matrix and labels from the REAL cluster_images on that S
                         (`--only-pagesim-bf16`)
  last_pooling.npz       REAL embedder.last_pooling on a seeded tensor
  vit_cases.npz          transformers.ViTModel (seeded synthetic weights) hidden states / embeddings
  crops/*.png            a few bundled region crops (data) incl. the 16 crops of config C1
  crops_expected.npz     Pillow resize + MllamaImageProcessorPil outputs for those crops
  tile_vit_cases.npz     transformers MllamaVisionModel (560 / 14 / 1280-d / 32 + 8 layers, seeded weights) rows per tile grid
                         (run separately: `make_golden.py --only-tile-vit`, ~15 min of CPU)
  query_cases.json       a brute-force store queried through the REAL safe_query (wrc:73-95): ids / distances
  nms_cases.json         boxes of the bundled pages (3_combined_bboxes/json, data files) and seeded box sets through the REAL
                         apply_non_max_suppression (3_combine_grids.py:80-137): kept indices in output order
"""
from __future__ import annotations

import glob
import hashlib
import html
import json
import os
import re
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
PKG = os.path.join(REF, "deprecated_package")
sys.path.insert(0, REPO)

os.environ.setdefault("MPLBACKEND", "Agg")


def import_reference():
    import transformers  # noqa: F401  (before stubbing torchvision)

    sys.path.insert(0, PKG)
    for n in [
        "chromadb",
        "chromadb.config",
        "chromadb.utils",
        "chromadb.utils.embedding_functions",
        "torchvision",
        "cv2",
        "pytesseract",
        "imutils",
    ]:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["chromadb.config"].Settings = object
    sys.modules["chromadb.utils.embedding_functions"].DefaultEmbeddingFunction = object
    sys.modules["pytesseract"].Output = object
    import embedder
    import weighted_region_clustering as w

    # per-pair JSON progress rewrites are O(P^4) bytes; keep them in memory
    done = set()
    w.is_clustering_completed = lambda key: False
    w.mark_clustering_as_completed = lambda key: done.add(key)
    return w, embedder


class FakeCollection:
    """Duck-typed exact-kNN stand-in for the chroma collection."""

    def __init__(self, ids, emb, metas, metric, renormalise=True):
        self.ids, self.emb, self.metas, self.metric = ids, np.asarray(emb, dtype=np.float64), metas, metric
        # renormalise=False: the rows are taken as stored (bf16-rounded unit rows: what the GPU table holds) and the
        # "cosine" is their plain inner product -- north_star's definition of the compare step (an MFMA GEMM over
        # L2-normalised rows), so the REAL function sees exactly the numbers the device kernel is asked to produce
        self.renormalise = renormalise
        self.norm = self.emb / np.linalg.norm(self.emb, axis=1, keepdims=True) if renormalise else self.emb

    def get(self, include=None, where=None, ids=None):
        return {"ids": list(self.ids), "metadatas": list(self.metas), "embeddings": [e.tolist() for e in self.emb]}

    def query(self, query_embeddings, n_results, include=None, where=None):
        q = np.asarray(query_embeddings[0], dtype=np.float64)
        if self.renormalise:
            q = q / np.linalg.norm(q)
        (key, cond), = where.items()
        rows = np.array([i for i, m in enumerate(self.metas) if m.get(key) == cond["$eq"]], dtype=np.int64)
        cos = self.norm[rows] @ q
        d = 1.0 - cos if self.metric == "cosine" else 2.0 - 2.0 * cos
        order = np.argsort(d, kind="stable")[:n_results]
        return {
            "ids": [[self.ids[rows[k]] for k in order]],
            "distances": [[float(d[k]) for k in order]],
            "metadatas": [[self.metas[rows[k]] for k in order]],
            "documents": [[None for _ in order]],
        }


def parse_report():
    path = os.path.join(PKG, "output/weighted_clustering/html_report/index.html")
    txt = open(path, encoding="utf-8").read()
    lines = txt.split("\n")
    k = int(re.search(r"Number of clusters: (\d+)", txt).group(1))
    # clusters
    membership = {}
    for m in re.finditer(r"<h3>Cluster (\d+)</h3>(.*?)</table>", txt, flags=re.S):
        lab = int(m.group(1))
        for nm in re.findall(r"<td>(.*?)</td>", m.group(2)):
            membership[html.unescape(nm)] = lab
    names = sorted(membership)
    row = lines[602]
    cells = re.findall(r"<td[^>]*>([0-9.]+)</td>", row)
    P = len(names)
    assert len(cells) == P * P, (len(cells), P)
    M = np.array([float(c) for c in cells]).reshape(P, P)
    labels = [membership[n] for n in names]
    top = []
    sec = txt[txt.index("<h2>Top Similarities</h2>") :]
    sec = sec[: sec.index("</table>")]
    for a, b, v in re.findall(r"<td>(.*?)</td>\s*<td>(.*?)</td>\s*<td>([0-9.]+)</td>", sec, flags=re.S):
        top.append([names.index(html.unescape(a)), names.index(html.unescape(b)), float(v)])
    return {"names": names, "matrix": M.tolist(), "labels": labels, "n_clusters": k, "top_pairs": top}


def load_region_table():
    pages = []
    for f in sorted(glob.glob(os.path.join(PKG, "output/region_cache/*.json"))):
        d = json.load(open(f))
        stem = os.path.basename(f).replace("_conf0.1_iou0.45.json", "")
        pages.append(
            {
                "name": stem + ".png",
                "width": d["image_size"]["width"],
                "height": d["image_size"]["height"],
                "boxes": d["boxes"],
                "classes": d["classes"],
                "scores": d["scores"],
                "class_names": d["class_names"],
            }
        )
    pages.sort(key=lambda p: p["name"])
    return pages


def table_rows(pages):
    """Rows as region_processor.py:75-113 would upsert them (type filter, int box, area %)."""
    from oracle.compare import REGION_TYPES_TO_PROCESS

    ids, metas, page_of = [], [], []
    for pi, p in enumerate(pages):
        for i, (box, cname) in enumerate(zip(p["boxes"], p["class_names"])):
            if cname not in REGION_TYPES_TO_PROCESS:
                continue
            x0, y0, x1, y1 = map(int, box)
            tot = p["width"] * p["height"]
            ap = ((x1 - x0) * (y1 - y0) / tot) * 100 if tot else 0
            ids.append(f"region_{os.path.splitext(p['name'])[0]}_{i}")
            metas.append({"parent_image_name": p["name"], "region_type": cname, "area_percentage": ap, "is_region": True})
            page_of.append(pi)
    return ids, metas, np.array(page_of)


def unit_vectors(n, d, seed, clusters=0):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, d))
    if clusters:
        c = rng.standard_normal((clusters, d)) * 2.0
        v = v + c[rng.integers(0, clusters, n)]
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)


class StoreFake:
    """Exact-kNN stand-in with the calls region_compare.py / cross_compare.py make: get(where),
    get(ids), query (cosine space: distance = 1 - cos, ascending, stable)."""

    def __init__(self, ids, emb, metas, docs):
        self.ids, self.metas, self.docs = list(ids), list(metas), list(docs)
        self.emb = np.asarray(emb, dtype=np.float64)
        self.norm = self.emb / np.linalg.norm(self.emb, axis=1, keepdims=True)
        self.pos = {i: k for k, i in enumerate(self.ids)}

    def _rows(self, where):
        if not where:
            return list(range(len(self.ids)))
        (key, cond), = where.items()
        return [i for i, m in enumerate(self.metas) if m.get(key) == cond["$eq"]]

    def get(self, ids=None, include=None, where=None):
        rows = self._rows(where) if ids is None else [self.pos[i] for i in ids if i in self.pos]
        return {"ids": [self.ids[r] for r in rows], "metadatas": [self.metas[r] for r in rows],
                "embeddings": [self.emb[r].tolist() for r in rows], "documents": [self.docs[r] for r in rows]}

    def query(self, query_embeddings, n_results, include=None, where=None):
        q = np.asarray(query_embeddings[0], dtype=np.float64)
        q = q / np.linalg.norm(q)
        rows = np.array(self._rows(where), dtype=np.int64)
        d = 1.0 - self.norm[rows] @ q
        order = np.argsort(d, kind="stable")[:n_results]
        return {"ids": [[self.ids[rows[k]] for k in order]], "distances": [[float(d[k]) for k in order]],
                "metadatas": [[self.metas[rows[k]] for k in order]], "documents": [[self.docs[rows[k]] for k in order]]}


def golden_neighbours():
    """What create_region_cross_comparison (region_compare.py:25) and create_cross_comparison
    (cross_compare.py:19) select, run on brute-force stores -> neighbour_cases.npz / .json."""
    import cross_compare as cc
    import region_compare as rc

    out = {}
    # -- regions: 9 pages, clustered 32-d vectors, exact duplicates, zero areas, a row without parent
    rng = np.random.default_rng(41)
    counts = [14, 3, 25, 11, 0, 1, 40, 12, 7]
    ids, metas, docs, page = [], [], [], []
    for p, c in enumerate(counts):
        for r in range(c):
            k = len(ids)
            ap = float(np.exp(rng.uniform(np.log(1e-2), np.log(20.0)))) if rng.random() > 0.08 else 0.0
            ids.append(f"region_p{p}_{r}")
            metas.append({"parent_image": f"/data/pages/Paper {p:02d}.png", "region_type": ["plain_text", "title", "figure"][k % 3],
                          "box_str": f"{k},0,{k + 1},1", "region_index": r, "area_percentage": ap, "is_region": True})
            docs.append(f"Region: plain_text from Paper {p:02d}.png")
            page.append(p)
    emb = unit_vectors(len(ids), 32, 43, clusters=6)
    emb[20] = emb[21]
    emb[70] = emb[71] = emb[72]
    emb[5] = emb[60]  # duplicates across pages: distance 0 < 0.3 -> dropped by the literal G2 rule
    del metas[33]["box_str"]  # a source the reference skips (:147-149) but still returns as a candidate
    store = StoreFake(ids, emb, metas, docs)
    picked = {}

    def record(parent_image, box, similar_parent, similar_box, score, vis_path):
        picked.setdefault(int(box[0]), []).append((int(similar_box[0]) if similar_box[2] else 33, float(score)))

    rc.create_region_comparison_visualization = record
    rc.is_region_comparison_completed = lambda rid: False
    rc.mark_region_comparison_as_completed = lambda rid: None
    rc.load_region_comparison_progress = lambda: {"completed_comparisons": []}
    rc.save_region_comparison_progress = lambda x: None
    rc.tqdm = lambda it, **kw: it
    assert rc.create_region_cross_comparison(store) is True
    n, top = len(ids), rc.REGION_COMPARE_TOP_N
    idx = np.full((n, top), -1, dtype=np.int64)
    score = np.zeros((n, top))
    for r, lst in picked.items():
        for k, (c, sc) in enumerate(lst):
            idx[r, k], score[r, k] = c, sc
    out.update(region_emb=emb, region_page=np.array(page), region_area=np.array([m["area_percentage"] for m in metas]),
               region_idx=idx, region_distance=score, region_no_box=np.array([33]))
    print("region neighbour golden:", int((idx >= 0).sum()), "picks over", n, "regions")

    # -- whole images: prefix rule of cross_compare.py:109-110,200-206
    os.makedirs("imgs", exist_ok=True)
    names = []
    for t, title in enumerate(["Addison NY Advertiser 1883", "Addison NY Advertiser 1884", "Bath Plaindealer 1885", "Corning Journal"]):
        for k in range(7 if t < 3 else 4):
            names.append(f"{title} - {k:04d}.png")
    names += ["a.png", "ab.png", "Short.png"]
    paths = [os.path.join(os.getcwd(), "imgs", nm) for nm in names]
    for pth in paths:
        open(pth, "w").close()  # candidates must exist on disk (:188-191)
    iemb = unit_vectors(len(names), 64, 47, clusters=4)
    iemb[3] = iemb[9]
    istore = StoreFake([f"image_{nm}" for nm in names], iemb, [{"image_path": pth} for pth in paths], [None] * len(names))
    cc.is_cross_compare_completed = lambda pth: False
    cc.mark_cross_compare_as_completed = lambda pth: None
    cc.load_cross_compare_progress = lambda: {"completed_images": []}
    cc.save_cross_compare_progress = lambda x: None
    cc.create_cross_comparison(None, istore, paths)
    top = cc.CROSS_COMPARE_TOP_N
    iidx = np.full((len(names), top), -1, dtype=np.int64)
    iscore = np.zeros((len(names), top))
    for r, nm in enumerate(names):
        stem = os.path.splitext(nm)[0].replace(" ", "_").replace(".", "_")
        page_html = open(os.path.join(cc.CROSS_COMPARE_FOLDER, "html_pages", stem + ".html")).read()
        hits = re.findall(r"<p><strong>(\d+)\.</strong> (.*?)</p>.*?Similarity score: <span class=\"score\">([0-9.\-]+)</span>", page_html, re.S)
        for k, (_, fname, sc) in enumerate(hits):
            iidx[r, k], iscore[r, k] = names.index(html.unescape(fname)), float(sc)
    out.update(image_emb=iemb, image_idx=iidx, image_distance_4dp=iscore)
    json.dump({"image_names": names}, open(os.path.join(HERE, "neighbour_names.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "neighbour_cases.npz"), **out)
    print("image neighbour golden:", int((iidx >= 0).sum()), "picks over", len(names), "images")


def golden_query(w):
    """`collection.query` results as the reference's own helper obtains them: the REAL `safe_query` (wrc:73-95) driving a
    brute-force store with the three `where` shapes of the call sites (wrc:204-208 parent page, region_compare.py:165-170
    is_region, cross_compare.py:119-123 none) -> query_cases.json (inputs + expected ids / distances)."""
    rng = np.random.default_rng(53)
    pages = [f"Paper {p:02d}.png" for p in range(6)]
    ids, metas, docs = [], [], []
    for k in range(180):
        p = int(rng.integers(0, 6))
        region = k % 9 != 0
        ids.append(f"{'region' if region else 'image'}_{k}")
        metas.append({"parent_image_name": pages[p], "is_region": region, "region_type": ["plain_text", "title", "figure"][k % 3],
                      "area_percentage": float(rng.uniform(0.1, 20.0))})
        docs.append(f"doc {k}")
    emb = unit_vectors(len(ids), 64, 59, clusters=5)
    emb[11] = emb[12] = emb[100]  # exact ties: insertion order decides
    store = StoreFake(ids, emb, metas, docs)
    queries = unit_vectors(7, 64, 61, clusters=5)
    queries[3] = emb[100]  # a stored vector as the query: distance 0 to its copies
    cases = []
    for qi, (where, n) in enumerate([({"parent_image_name": {"$eq": pages[2]}}, 10), ({"is_region": {"$eq": True}}, 30), (None, 25),
                                     ({"is_region": {"$eq": True}}, 15), ({"parent_image_name": {"$eq": pages[5]}}, 100),
                                     ({"parent_image_name": {"$eq": "absent.png"}}, 10), (None, 150)]):
        res = w.safe_query(store, queries[qi].tolist(), n, where)
        cases.append({"query": qi, "where": where, "n_results": n, "ids": res["ids"][0], "distances": res["distances"][0]})
    json.dump({"ids": ids, "metadatas": metas, "documents": docs, "embeddings": emb.tolist(), "queries": queries.tolist(), "cases": cases},
              open(os.path.join(HERE, "query_cases.json"), "w"))
    print("query golden:", [len(c["ids"]) for c in cases], "results")


def golden_regions():
    """RegionProcessor.process_image_regions (region_processor.py:62) + the real
    DocLayoutDetector.get_region_image (doclayout_detector.py:165) on a seeded page -> region_rows.json:
    ids, metadata rows, documents and the sha256 of every crop the embedder was handed."""
    from PIL import Image

    for n in ["doclayout_yolo", "ultralytics"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["doclayout_yolo"].YOLOv10 = object
    import region_processor as rp
    from doclayout_detector import DocLayoutDetector

    seed, H, W = 77, 220, 300
    page = np.random.default_rng(seed).integers(0, 256, (H, W, 3), dtype=np.uint8)
    os.makedirs("pages", exist_ok=True)
    page_path = os.path.join(os.getcwd(), "pages", "Seeded Gazette 1901 - 0007.png")
    Image.fromarray(page).save(page_path)
    regions = {
        "boxes": [[10.7, 20.2, 99.9, 80.5], [0.0, 0.0, 300.0, 220.0], [150.2, 5.9, 299.99, 40.1], [-3.6, 100.4, 40.2, 130.9],
                  [250.5, 180.3, 320.7, 240.2], [30.0, 30.0, 31.9, 200.0], [5.5, 210.1, 295.5, 219.9], [60.0, 60.0, 120.0, 90.0]],
        "classes": [1, 0, 3, 1, 5, 1, 4, 2],
        "class_names": ["plain_text", "title", "figure", "plain_text", "table", "plain_text", "figure_caption", "abandon"],
        "scores": [0.91, 0.88, 0.75, 0.66, 0.5, 0.41, 0.33, 0.2],
        "image_size": {"width": W, "height": H},
    }
    crops, upserts = [], []

    class Emb:
        def get_image_embeddings(self, paths):
            for pth in paths:
                crops.append(np.array(Image.open(pth).convert("RGB")))  # embedder.py:107
            return [[float(k)] * 4 for k in range(len(paths))]

    class Col:
        def upsert(self, ids, embeddings, documents, metadatas):
            upserts.append((ids, documents, metadatas))

    det = types.SimpleNamespace(get_region_image=lambda pth, box, padding=0: DocLayoutDetector.get_region_image(None, pth, box, padding))
    rp.is_region_embedding_completed = lambda rid: False
    rp.mark_region_embedding_as_completed = lambda rid: None
    count = rp.RegionProcessor(Emb(), Col(), det).process_image_regions(page_path, regions)
    ids = [i for u in upserts for i in u[0]]
    docs = [d for u in upserts for d in u[1]]
    metas = [dict(m, parent_image="<page_path>") for u in upserts for m in u[2]]
    assert count == len(ids) == len(crops)
    json.dump({"seed": seed, "page_hw": [H, W], "page_name": os.path.basename(page_path), "regions": regions, "ids": ids,
               "documents": docs, "metadatas": metas, "crop_shapes": [list(c.shape) for c in crops],
               "crop_sha256": [hashlib.sha256(np.ascontiguousarray(c).tobytes()).hexdigest() for c in crops]},
              open(os.path.join(HERE, "region_rows.json"), "w"), indent=1)
    print("region rows golden:", count, "regions;", [c.shape for c in crops])


TILE_CASE_SIZES = [(50, 40), (300, 1500), (1500, 300), (600, 600), (1200, 1100), (561, 1130), (2000, 500), (500, 2000), (560, 560),
                   (7, 5), (559, 561), (1121, 560), (2240, 560), (560, 2240), (3000, 3000), (70, 1200), (1681, 560), (5114, 63),
                   (20, 3862), (1119, 1121), (1000, 500), (450, 900)]


def golden_tiles():
    """transformers MllamaImageProcessorPil with the checkpoint's geometry (tile 560, <= 4 tiles, CLIP mean/std)
    on the committed crops and on seeded arrays covering all eight tile arrangements -> tile_cases.json
    (aspect-ratio ids, tile counts, sha256 + probes of the f32 pixel_values)."""
    from PIL import Image
    from transformers.models.mllama.image_processing_pil_mllama import MllamaImageProcessorPil

    from oracle.preprocess import CLIP_MEAN, CLIP_STD

    proc = MllamaImageProcessorPil(size={"height": 560, "width": 560}, max_image_tiles=4, image_mean=list(CLIP_MEAN), image_std=list(CLIP_STD))
    man = json.load(open(os.path.join(HERE, "crops_manifest.json")))
    cases = []
    images = [("file:" + c["file"], np.array(Image.open(os.path.join(HERE, "crops", c["file"])).convert("RGB"))) for c in man["crops"]]
    for k, (h, w) in enumerate(TILE_CASE_SIZES):
        images.append((f"seed:{1000 + k}:{h}x{w}", np.random.default_rng(1000 + k).integers(0, 256, (h, w, 3), dtype=np.uint8)))
    probe = np.random.default_rng(5).integers(0, 4 * 3 * 560 * 560, 24)
    for name, img in images:
        out = proc(images=[img], return_tensors="np")
        pv = np.ascontiguousarray(out["pixel_values"][0, 0])
        cases.append({"source": name, "hw": list(img.shape[:2]), "aspect_ratio_id": int(out["aspect_ratio_ids"][0, 0]),
                      "aspect_ratio_mask": out["aspect_ratio_mask"][0, 0].tolist(), "num_tiles": int(out["num_tiles"][0][0]),
                      "sha256": hashlib.sha256(pv.tobytes()).hexdigest(), "probe": [float(v) for v in pv.reshape(-1)[probe]]})
    json.dump({"tile": 560, "max_tiles": 4, "probe_index": probe.tolist(), "cases": cases}, open(os.path.join(HERE, "tile_cases.json"), "w"))
    print("tile golden:", len(cases), "crops; arrangements", sorted({c["aspect_ratio_id"] for c in cases}))


def golden_tile_vit():
    """transformers `MllamaVisionModel(MllamaVisionConfig(image_size=560))` -- the reference encoder's own vision tower
    (embedder.py:75-79,117-126; 32 local + 8 gated global layers, 1280-d, <= 4 tiles of 1601 tokens, 7680-d output) --
    holding the seeded synthetic weights of `make_tile_vit_weights(2)`, fp32 on the CPU, on one image per tile
    arrangement (the first case of every aspect-ratio id in tile_cases.json, through transformers' own image
    processor) -> tile_vit_cases.npz: per case the rows of tokens {0, 1, 800, 1600} of every real tile (f16) and the
    oracle restatement's agreement with them."""
    import time

    import torch
    from PIL import Image
    from transformers.models.mllama.configuration_mllama import MllamaVisionConfig
    from transformers.models.mllama.image_processing_pil_mllama import MllamaImageProcessorPil
    from transformers.models.mllama.modeling_mllama import MllamaVisionModel

    from multimodal_embeddings_amd.weights import make_tile_vit_weights
    from oracle import mllama_vision as om
    from oracle.preprocess import CLIP_MEAN, CLIP_STD

    torch.set_num_threads(8)
    w = make_tile_vit_weights(2)
    model = MllamaVisionModel(MllamaVisionConfig(image_size=560)).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    proc = MllamaImageProcessorPil(size={"height": 560, "width": 560}, max_image_tiles=4, image_mean=list(CLIP_MEAN), image_std=list(CLIP_STD))
    g = json.load(open(os.path.join(HERE, "tile_cases.json")))
    first = {}
    for c in g["cases"]:
        if c["source"].startswith("seed:"):
            first.setdefault(c["aspect_ratio_id"], c)
    tokens = [0, 1, 800, 1600]
    out = {"tokens": np.array(tokens)}
    sources = []
    for aid in sorted(first):
        c = first[aid]
        seed, hw = c["source"].split(":")[1:]
        h, wd = map(int, hw.split("x"))
        img = np.random.default_rng(int(seed)).integers(0, 256, (h, wd, 3), dtype=np.uint8)
        inp = proc(images=[img], return_tensors="pt")
        assert int(inp["aspect_ratio_ids"][0, 0]) == aid
        t0 = time.time()
        with torch.no_grad():
            hs = model(pixel_values=inp["pixel_values"], aspect_ratio_ids=inp["aspect_ratio_ids"],
                       aspect_ratio_mask=inp["aspect_ratio_mask"]).last_hidden_state[0, 0].numpy()
        nt = c["num_tiles"]
        rows = hs[:nt][:, tokens]  # [nt, 4, 7680]
        out[f"rows_{aid}"] = rows.astype(np.float16)
        sources.append(c["source"])
        print(f"tile-ViT golden: aspect id {aid} ({nt} tiles, {c['source']}) in {time.time() - t0:.0f} s, |rows| {np.abs(rows).max():.3f}", flush=True)
        if aid in (1, 6):  # the oracle restatement against the real class: one single-tile and one four-tile image
            t0 = time.time()
            got = om.vision_forward(inp["pixel_values"][0, 0].numpy(), aid, nt, w)
            err = float(np.abs(got[:nt] - hs[:nt]).max() / np.abs(hs[:nt]).max())
            cos = (got[:nt] * hs[:nt]).sum(-1) / np.linalg.norm(got[:nt], axis=-1) / np.linalg.norm(hs[:nt], axis=-1)
            print(f"  oracle vs transformers: max rel err {err:.2e}, min token cosine {cos.min():.8f} ({time.time() - t0:.0f} s)", flush=True)
            assert err < 1e-4 and cos.min() > 1 - 1e-6
            out[f"oracle_max_rel_err_{aid}"] = np.array(err)
    out["sources"] = np.array(sources)
    np.savez_compressed(os.path.join(HERE, "tile_vit_cases.npz"), **out)
    print("tile-ViT golden written:", sorted(first))


def golden_nms():
    """The REAL apply_non_max_suppression (3_combine_grids.py:80-137) on the bundled pages' combined boxes at several
    thresholds and on seeded box sets (ties, duplicates, touching and degenerate boxes): kept indices in output order.
    The function returns boxes, not indices; the inner box lists keep their identity through its shallow copies."""
    import importlib.util

    for n in ["cv2"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    spec = importlib.util.spec_from_file_location("ref_combine_grids", os.path.join(REF, "3_combine_grids.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def run(boxes, scores, classes, thr):
        boxes = [list(map(float, b)) for b in boxes]
        ident = {id(b): i for i, b in enumerate(boxes)}
        names = [f"class{c}" for c in classes]
        fb, fs, fc, fn = mod.apply_non_max_suppression(boxes, list(scores), list(classes), names, thr)
        keep = [ident[id(b)] for b in fb]
        assert [scores[i] for i in keep] == fs and [classes[i] for i in keep] == fc and [names[i] for i in keep] == fn
        return keep

    cases = []
    pages = sorted(glob.glob(os.path.join(REF, "3_combined_bboxes", "json", "*_combined.json")))
    for path in pages:
        d = json.load(open(path))
        case = {"name": os.path.basename(path)[:40], "boxes": d["boxes"], "scores": d["scores"], "classes": d["classes"], "runs": []}
        for thr in (0.5, 0.3, 0.1, 0.0):
            case["runs"].append({"iou_threshold": thr, "keep": run(d["boxes"], d["scores"], d["classes"], thr)})
        # the bundled files ARE the reference's output at 0.5: suppression must be idempotent on them
        assert case["runs"][0]["keep"] == sorted(range(len(d["scores"])), key=lambda i: -d["scores"][i]), path
        cases.append(case)
    rng = np.random.default_rng(31)
    for n, kind in [(1, "one"), (2, "pair"), (7, "few"), (64, "int"), (300, "mixed"), (1200, "dense"), (257, "ties")]:
        centres = rng.uniform(0, 2000, (max(n // 6, 1), 2))
        c = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 12, (n, 2))
        wh = rng.uniform(20, 400, (n, 2))
        boxes = np.concatenate([c - wh / 2, c + wh / 2], axis=1)
        if kind in ("int", "ties"):
            boxes = np.round(boxes)  # whole-pixel boxes: exact touching edges and equal IoUs
        scores = rng.uniform(0.2, 1.0, n)
        if kind in ("ties", "int"):
            scores = np.round(scores, 1)  # many equal scores: the first in list order wins
        classes = rng.integers(0, 4 if kind != "dense" else 10, n)
        if n >= 7:
            boxes[3] = boxes[1]                                   # exact duplicate
            classes[3] = classes[1]
            boxes[5] = [boxes[4][2], boxes[4][1], boxes[4][2] + 50, boxes[4][3]]  # shares an edge with box 4 (area branch, IoU 0)
            classes[5] = classes[4]
            boxes[6] = [100.0, 100.0, 100.0, 300.0]               # zero width
        boxes, scores, classes = boxes.tolist(), scores.tolist(), [int(v) for v in classes]
        case = {"name": f"seeded {kind} n={n}", "boxes": boxes, "scores": scores, "classes": classes, "runs": []}
        for thr in (0.5, 0.25, 0.0):
            case["runs"].append({"iou_threshold": thr, "keep": run(boxes, scores, classes, thr)})
        cases.append(case)
    cases.append({"name": "empty", "boxes": [], "scores": [], "classes": [], "runs": [{"iou_threshold": 0.5, "keep": []}]})
    json.dump({"cases": cases}, open(os.path.join(HERE, "nms_cases.json"), "w"))
    print("nms:", len(cases), "cases,", sum(len(c["scores"]) for c in cases), "boxes")


def bf16_round(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def golden_pagesim_bf16(w):
    """pagesim_bf16_cases.npz: the REAL compute_image_similarity_matrix (wrc:97-254) and the REAL cluster_images
    (wrc:452-574) fed the bf16-ROUNDED unit rows the device table holds: the GPU page matrix is then
    compared with the reference's own output directly, not through the oracle and not through a correlation."""
    pages = json.load(open(os.path.join(HERE, "region_table.json")))
    ids, metas, page_of = table_rows(pages)
    names19 = [p["name"] for p in pages]
    paths19 = ["/somewhere/" + n for n in names19]
    pc = {}
    emb16 = bf16_round(unit_vectors(len(ids), 64, 7, clusters=12))  # same seeded rows as pagesim_cases.npz, rounded
    pc["real_emb_bf16"] = emb16
    pc["real_area_percentage"] = np.array([m["area_percentage"] for m in metas])
    pc["real_page_of"] = page_of
    for metric in ("cosine", "sqeuclidean"):
        S, nm = w.compute_image_similarity_matrix(FakeCollection(ids, emb16, metas, metric, renormalise=False), paths19)
        assert nm == names19
        pc[f"real_S_{metric}"] = S
        res = w.cluster_images(S.copy(), list(nm))
        pc[f"real_labels_{metric}"] = np.array(res["labels"])
        pc[f"real_k_{metric}"] = np.array(res["n_clusters"])
    # a larger synthetic table (P = 48 pages, 12..90 regions, D = 192): more top-k boundaries
    rng = np.random.default_rng(4242)
    P = 48
    syn_names = [f"Synthetic Gazette {i:03d} of the bf16 fixture.png" for i in range(P)]
    counts = rng.integers(12, 91, P)
    s_ids, s_metas, s_page = [], [], []
    for p, c in enumerate(counts):
        for r in range(int(c)):
            ap = float(np.exp(rng.uniform(np.log(1e-2), np.log(20.0))))
            s_ids.append(f"region_p{p}_{r}")
            s_metas.append({"parent_image_name": syn_names[p], "region_type": "plain_text", "area_percentage": ap, "is_region": True})
            s_page.append(p)
    s_emb = bf16_round(unit_vectors(len(s_ids), 192, 11, clusters=9))
    pc["syn_emb_bf16"] = s_emb.astype(np.float32)
    pc["syn_area_percentage"] = np.array([m["area_percentage"] for m in s_metas])
    pc["syn_page_of"] = np.array(s_page)
    S, nm = w.compute_image_similarity_matrix(FakeCollection(s_ids, s_emb, s_metas, "cosine", renormalise=False), ["/x/" + n for n in syn_names])
    pc["syn_S_cosine"] = S
    res = w.cluster_images(S.copy(), list(nm))
    pc["syn_labels_cosine"] = np.array(res["labels"])
    pc["syn_k_cosine"] = np.array(res["n_clusters"])
    json.dump({"names": syn_names}, open(os.path.join(HERE, "pagesim_bf16_names.json"), "w"))
    # bf16 rows are stored as their 16-bit patterns (half the bytes, exact)
    for k in ("real_emb_bf16", "syn_emb_bf16"):
        pc[k] = (pc[k].astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, "pagesim_bf16_cases.npz"), **pc)
    print("pagesim bf16 cases ok: labels", pc["real_labels_cosine"].tolist(), "k", int(pc["real_k_cosine"]), "| syn k", int(pc["syn_k_cosine"]))


def main():
    scratch = tempfile.mkdtemp(prefix="golden_")
    os.chdir(scratch)
    os.makedirs("output", exist_ok=True)
    w, ref_embedder = import_reference()
    if "--only-tile-vit" in sys.argv:
        golden_tile_vit()
        shutil.rmtree(scratch, ignore_errors=True)
        return
    if "--only-pagesim-bf16" in sys.argv:
        golden_pagesim_bf16(w)
        shutil.rmtree(scratch, ignore_errors=True)
        return
    if "--only-nms" in sys.argv:
        golden_nms()
        shutil.rmtree(scratch, ignore_errors=True)
        return
    if "--only-query" in sys.argv:
        golden_query(w)
        shutil.rmtree(scratch, ignore_errors=True)
        return
    if "--only-next" in sys.argv:  # the SURVEY 8(f) rows only (leaves the other fixtures untouched)
        golden_neighbours()
        golden_regions()
        golden_tiles()
        shutil.rmtree(scratch, ignore_errors=True)
        return
    import torch

    # ---- 1. report KAT -------------------------------------------------------------
    rep = parse_report()
    M = np.array(rep["matrix"])
    res = w.cluster_images(M.copy(), list(rep["names"]))
    assert res["labels"] == rep["labels"], (res["labels"], rep["labels"])
    assert res["n_clusters"] == rep["n_clusters"]
    rep["cohesion"] = {str(k): v for k, v in res["cluster_cohesion"].items()}
    json.dump(rep, open(os.path.join(HERE, "report_matrix.json"), "w"))
    print("report KAT ok: labels", res["labels"])

    # ---- 2. region table ----------------------------------------------------------
    pages = load_region_table()
    json.dump(pages, open(os.path.join(HERE, "region_table.json"), "w"), separators=(",", ":"))
    ids, metas, page_of = table_rows(pages)
    assert len(ids) == 1867, len(ids)
    progress = json.load(open(os.path.join(PKG, "output/region_embedding_progress.json")))
    key = next(iter(progress))
    assert sorted(progress[key]) == sorted(ids), "region ids differ from the reference's progress file"

    # ---- 3. cluster_images on seeded matrices ---------------------------------------
    from scipy.cluster import hierarchy
    from sklearn.cluster import AgglomerativeClustering
    from sklearn.metrics import silhouette_score

    cc, lc = {}, {}
    rng = np.random.default_rng(1234)
    case = 0
    for P, kind in [(2, "dense"), (3, "dense"), (5, "dense"), (8, "sparse"), (10, "dense"), (12, "blocks"), (19, "sparse"), (40, "blocks"), (64, "dense"), (7, "few")]:
        A = rng.random((P, P))
        S = (A + A.T) / 2
        if kind == "sparse":
            S[rng.random((P, P)) < 0.6] = 0
            S = np.maximum(S, S.T) * (S > 0) * (S.T > 0)
        if kind == "blocks":
            g = rng.integers(0, 4, P)
            S = S * 0.2 + 0.7 * (g[:, None] == g[None, :])
        if kind == "few":
            S = np.zeros((P, P))
            S[0, 1] = S[1, 0] = 1.0
            S[2, 3] = S[3, 2] = 0.4
        mx = np.max(S - np.diag(np.diag(S)))
        S = S / mx
        np.fill_diagonal(S, 1.0)
        names = [f"page_{i:03d}.png" for i in range(P)]
        for fixed in (None, 3 if P >= 3 else 2):
            r = w.cluster_images(S.copy(), names, n_clusters=fixed)
            if r is None:
                # the reference returns None when sklearn raises (e.g. k > P); record that too
                cc[f"c{case}_S"] = S
                cc[f"c{case}_fixed"] = np.array(-1 if fixed is None else fixed)
                cc[f"c{case}_none"] = np.array(1)
                case += 1
                continue
            cc[f"c{case}_S"] = S
            cc[f"c{case}_fixed"] = np.array(-1 if fixed is None else fixed)
            cc[f"c{case}_labels"] = np.array(r["labels"])
            cc[f"c{case}_k"] = np.array(r["n_clusters"])
            cc[f"c{case}_coh_keys"] = np.array(sorted(r["cluster_cohesion"]))
            cc[f"c{case}_coh_vals"] = np.array([r["cluster_cohesion"][k] for k in sorted(r["cluster_cohesion"])])
            case += 1
        if P >= 3:
            D = 1.0 - S
            Z = hierarchy.linkage(D, method="average", metric="euclidean")
            lc[f"l{P}_{kind}_D"] = D
            lc[f"l{P}_{kind}_Z"] = Z
            Zp = hierarchy.linkage(D[np.triu_indices(P, 1)], method="average")
            lc[f"l{P}_{kind}_Zpre"] = Zp
            for k in range(2, min(10, P) + 1):
                lab = AgglomerativeClustering(n_clusters=k, linkage="average").fit(D).labels_
                lc[f"l{P}_{kind}_lab{k}"] = lab
                if 1 < len(np.unique(lab)) < P:
                    lc[f"l{P}_{kind}_sil{k}"] = np.array(silhouette_score(D, lab, metric="precomputed"))
                labp = AgglomerativeClustering(n_clusters=k, linkage="average", metric="precomputed").fit(D).labels_
                lc[f"l{P}_{kind}_labpre{k}"] = labp
    cc["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(HERE, "cluster_cases.npz"), **cc)
    np.savez_compressed(os.path.join(HERE, "linkage_cases.npz"), **lc)
    print("cluster cases:", case)

    # ---- 4. compute_image_similarity_matrix over a brute-force collection -----------
    pc = {}
    names19 = [p["name"] for p in pages]
    paths19 = ["/somewhere/" + n for n in names19]
    emb = unit_vectors(len(ids), 64, 7, clusters=12)
    pc["real_emb"] = emb
    pc["real_area_percentage"] = np.array([m["area_percentage"] for m in metas])
    pc["real_page_of"] = page_of
    for metric in ("cosine", "sqeuclidean"):
        S, nm = w.compute_image_similarity_matrix(FakeCollection(ids, emb, metas, metric), paths19)
        assert nm == names19
        pc[f"real_S_{metric}"] = S
    # the zero pattern of the bundled report must reappear (same-prefix skips)
    Z = np.array(rep["matrix"]) == 0
    assert np.array_equal(pc["real_S_cosine"][Z], np.zeros(Z.sum())), "same-prefix zero pattern differs"
    S_np, _ = w.compute_image_similarity_matrix(FakeCollection(ids, emb, metas, "cosine"), paths19, skip_same_prefix=False)
    pc["real_S_cosine_noskip"] = S_np

    # synthetic table with edge cases: empty page, zero-area rows, <10 regions, foreign types
    rng = np.random.default_rng(99)
    P = 9
    syn_names = [f"{'Same Prefix Newspaper Title':<20}{i}.png" if i in (2, 3) else f"Paper {i:02d} of the synthetic set.png" for i in range(P)]
    counts = [14, 3, 25, 11, 0, 1, 40, 12, 7]
    s_ids, s_metas, s_page = [], [], []
    for p, c in enumerate(counts):
        for r in range(c):
            ap = float(np.exp(rng.uniform(np.log(1e-2), np.log(20.0))))
            if rng.random() < 0.08:
                ap = 0.0
            typ = "plain_text" if rng.random() > 0.05 else "abandon"
            s_ids.append(f"region_p{p}_{r}")
            s_metas.append({"parent_image_name": syn_names[p], "region_type": typ, "area_percentage": ap, "is_region": True})
            s_page.append(p)
    s_emb = unit_vectors(len(s_ids), 32, 5, clusters=5)
    # exact duplicate vectors to exercise tie ordering
    s_emb[20] = s_emb[21]
    s_emb[70] = s_emb[71] = s_emb[72]
    pc["syn_emb"] = s_emb
    pc["syn_area_percentage"] = np.array([m["area_percentage"] for m in s_metas])
    pc["syn_page_of"] = np.array(s_page)
    pc["syn_types_ok"] = np.array([m["region_type"] == "plain_text" for m in s_metas])
    json.dump({"names": syn_names}, open(os.path.join(HERE, "pagesim_names.json"), "w"))
    for metric in ("cosine", "sqeuclidean"):
        S, nm = w.compute_image_similarity_matrix(FakeCollection(s_ids, s_emb, s_metas, metric), ["/x/" + n for n in syn_names])
        pc[f"syn_S_{metric}"] = S
    np.savez_compressed(os.path.join(HERE, "pagesim_cases.npz"), **pc)
    print("pagesim cases ok")
    golden_pagesim_bf16(w)
    golden_neighbours()
    golden_regions()
    golden_tiles()
    golden_query(w)

    # ---- 5. last_pooling -------------------------------------------------------------
    g = torch.Generator().manual_seed(3)
    hs = torch.randn(5, 9, 16, generator=g)
    mask = torch.tensor([[1] * 9, [1] * 4 + [0] * 5, [1] + [0] * 8, [1] * 7 + [0] * 2, [1] * 2 + [0] * 7])
    out = ref_embedder.last_pooling(hs, mask)
    out_raw = ref_embedder.last_pooling(hs, mask, normalize=False)
    np.savez_compressed(os.path.join(HERE, "last_pooling.npz"), hs=hs.numpy(), mask=mask.numpy(), out=out.numpy(), out_raw=out_raw.numpy())

    # ---- 6. transformers ViT with the seeded synthetic weights -------------------------
    from transformers import ViTConfig, ViTModel

    from multimodal_embeddings_amd.weights import make_vit_weights, synthetic_crops
    from oracle.preprocess import patchify, preprocess_crop
    from oracle.vit import vit_embed, vit_hidden_states

    vc = {}
    for tag, std in (("std002", 0.02), ("std008", 0.08)):
        wts = make_vit_weights(seed=1, std=std)
        model = ViTModel(ViTConfig(), add_pooling_layer=False).eval()
        sd = {k: torch.from_numpy(v.copy()) for k, v in wts.items()}
        missing = model.load_state_dict(sd, strict=True)
        crops = synthetic_crops(3, seed=0)
        px = np.stack([preprocess_crop(c) for c in crops])
        with torch.no_grad():
            hf = model(pixel_values=torch.from_numpy(px)).last_hidden_state.numpy()
        mine = vit_hidden_states(torch.from_numpy(np.stack([patchify(p) for p in px])), wts).numpy()
        err = np.abs(hf - mine).max()
        print(tag, "oracle vs transformers ViTModel max abs err", err)
        assert err < 2e-4, err
        cls = hf[:, 0] / np.linalg.norm(hf[:, 0], axis=1, keepdims=True)
        last = hf[:, -1] / np.linalg.norm(hf[:, -1], axis=1, keepdims=True)
        vc[f"{tag}_cls"] = cls.astype(np.float32)
        vc[f"{tag}_last"] = last.astype(np.float32)
        vc[f"{tag}_hidden_sample"] = hf[:, ::49, ::64].astype(np.float32)
    vc["crop_seed"] = np.array(0)
    vc["n"] = np.array(3)
    np.savez_compressed(os.path.join(HERE, "vit_cases.npz"), **vc)

    # ---- 7. real crops through Pillow + the Mllama image processor ----------------------
    from PIL import Image
    from transformers.models.mllama.image_processing_pil_mllama import MllamaImageProcessorPil

    from oracle.preprocess import CLIP_MEAN, CLIP_STD, fit_to_canvas

    crop_dir = os.path.join(PKG, "output/region_images")
    allf = sorted(os.listdir(crop_dir))
    sizes = {f: Image.open(os.path.join(crop_dir, f)).size for f in allf}
    c1_page = "Addision NY Advertiser 1883-1887 - 0001.pdf_387d113864_page_0000"
    c1 = sorted([f for f in allf if f.startswith(c1_page)], key=lambda f: int(re.search(r"_region(\d+)_", f).group(1)))[:16]
    small = [f for f in allf if os.path.getsize(os.path.join(crop_dir, f)) < 120_000]
    by_aspect = sorted(small, key=lambda f: sizes[f][0] / sizes[f][1])
    by_area = sorted(small, key=lambda f: sizes[f][0] * sizes[f][1])
    extra = [by_aspect[0], by_aspect[1], by_aspect[-1], by_aspect[-2], by_area[0], by_area[1], by_area[-1], by_aspect[len(by_aspect) // 2]]
    picked = c1 + [f for f in extra if f not in c1]
    out_dir = os.path.join(HERE, "crops")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    proc = MllamaImageProcessorPil(size={"height": 224, "width": 224}, max_image_tiles=1, image_mean=list(CLIP_MEAN), image_std=list(CLIP_STD))
    ce = {}
    manifest = []
    for i, f in enumerate(picked):
        shutil.copyfile(os.path.join(crop_dir, f), os.path.join(out_dir, f))
        os.chmod(os.path.join(out_dir, f), 0o644)
        im = Image.open(os.path.join(crop_dir, f))
        wd, ht = im.size
        nh, nw = fit_to_canvas(ht, wd)
        rs = np.array(im.convert("RGB").resize((nw, nh), resample=Image.BILINEAR))
        pv = proc(images=[im], return_tensors="np")["pixel_values"][0, 0, 0]
        ce[f"resized_{i}"] = rs
        ce[f"pv_sha256_{i}"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(pv).tobytes()).digest(), dtype=np.uint8)
        ce[f"pv_sample_{i}"] = pv[:, ::7, ::5].copy()
        manifest.append({"file": f, "width": wd, "height": ht, "new_h": nh, "new_w": nw})
    json.dump({"c1_count": len(c1), "crops": manifest}, open(os.path.join(HERE, "crops_manifest.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "crops_expected.npz"), **ce)
    # crop sizes of the whole bundled set (for the C3 workload shape): int-truncated bbox sizes
    all_sizes = np.array([[sizes[f][1], sizes[f][0]] for f in allf], dtype=np.int32)
    np.save(os.path.join(HERE, "bundled_crop_sizes_hw.npy"), all_sizes)
    print("crops:", len(picked), "fixtures written to", HERE)


if __name__ == "__main__":
    main()
