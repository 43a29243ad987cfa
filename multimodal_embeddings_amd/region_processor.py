"""Bounding boxes of a page -> region rows in the vector store (the step feeding the hot path).

Mirrors `RegionProcessor.process_image_regions` (deprecated_package/region_processor.py:62-158)
and the cache schema of `DocLayoutDetector.detect_regions` (doclayout_detector.py:145-153).  The
reference re-opens and re-decodes the full page PNG once PER REGION
(`get_region_image`, doclayout_detector.py:178), writes every crop to a PNG and re-reads it in the
embedder; here the page is decoded once, uploaded once, all boxes are cut on the device (K0
`mme_crop_boxes`) and go straight into the preprocessing + encoder kernels.
"""
from __future__ import annotations

import json
import logging
import os

import numpy as np

from . import config
from ._lib import MmeError

logger = logging.getLogger(__name__)


def load_region_cache(path):
    """One `region_cache/*.json` file (doclayout_detector.py:145-153): dict with `boxes`, `classes`,
    `scores`, `class_names`, `image_size {width, height}`."""
    with open(path) as fh:
        regions = json.load(fh)
    for key in ("boxes", "classes", "class_names", "scores"):
        regions.setdefault(key, [])
    regions.setdefault("image_size", {"width": 0, "height": 0})
    return regions


def region_rows(image_path, regions, region_types=None):
    """The rows region_processor.py:75-113 builds for one page, without touching pixels.

    Returns (ids, metadatas, int_boxes int32[n,4]) for the boxes whose class is embedded
    (:76-77), in detector order; `int_boxes` are the `map(int, box)` corners (:88) that define both
    the crop (doclayout_detector.py:179) and `area_percentage` (:89-93)."""
    region_types = config.REGION_TYPES_TO_PROCESS if region_types is None else region_types
    image_filename = os.path.basename(image_path)
    image_size = regions.get("image_size", {"width": 0, "height": 0})
    ids, metas, boxes = [], [], []
    for i, (box, class_id, class_name, score) in enumerate(
        zip(regions.get("boxes", []), regions.get("classes", []), regions.get("class_names", []), regions.get("scores", []))
    ):
        if class_name not in region_types:
            continue
        x_min, y_min, x_max, y_max = map(int, box)
        region_width, region_height = x_max - x_min, y_max - y_min
        total_area = image_size["width"] * image_size["height"]
        area_percentage = (region_width * region_height / total_area) * 100 if total_area else 0
        ids.append(f"region_{os.path.splitext(image_filename)[0]}_{i}")
        metas.append({
            "parent_image": image_path,
            "parent_image_name": image_filename,
            "region_index": i,
            "region_type": class_name,
            "region_class_id": int(class_id),
            "region_score": float(score),
            "box": ",".join(map(str, box)),
            "box_normalized": ",".join(map(str, [x_min / image_size["width"], y_min / image_size["height"],
                                                 x_max / image_size["width"], y_max / image_size["height"]])) if total_area else "",
            "area_percentage": area_percentage,
            "width": region_width,
            "height": region_height,
            "is_region": True,
        })
        boxes.append([x_min, y_min, x_max, y_max])
    return ids, metas, np.asarray(boxes, dtype=np.int32).reshape(-1, 4)


class RegionProcessor:
    """Same constructor shape and entry point as region_processor.py:RegionProcessor.

    `embedder` is a `RegionEmbedder`; `collection` anything with chroma's `upsert`.  `detector` is
    not needed: crops are cut on the GPU from the decoded page."""

    def __init__(self, embedder, collection, detector=None):
        self.embedder, self.collection, self.detector = embedder, collection, detector

    def embed_page_regions(self, page, int_boxes):
        """page: path | PIL image | uint8[H,W,3]; int_boxes int32[n,4] -> float32 CUDA tensor [n, 768].

        Boxes of zero or negative size (which make the reference's PNG save fail, :115-117) raise."""
        from .embedder import _load_rgb

        t = self.embedder.torch
        arr = _load_rgb(page)
        dev = t.device(f"cuda:{self.embedder.engine.device}")
        page_dev = t.from_numpy(np.require(arr, requirements=["C", "W"])).to(dev)
        pix, offs, hw = self.embedder.engine.crop_boxes(page_dev, int_boxes)
        e32, _ = self.embedder.embed_packed(pix, offs, hw, want_bf16=False)
        return e32

    def process_image_regions(self, image_path, regions, page=None):
        """region_processor.py:62-158: embed the page's regions and upsert them; returns the count.

        `page` overrides the pixels read from `image_path` (already decoded page)."""
        image_filename = os.path.basename(image_path)
        if not regions.get("boxes"):
            return 0
        ids, metas, boxes = region_rows(image_path, regions)
        if not ids:
            return 0
        good = [k for k in range(len(ids)) if boxes[k, 2] > boxes[k, 0] and boxes[k, 3] > boxes[k, 1]]
        for k in sorted(set(range(len(ids))) - set(good)):
            logger.warning(f"Failed to extract region {metas[k]['region_index']} from {image_filename}")  # :85-87
        if not good:
            return 0
        try:
            emb = self.embed_page_regions(image_path if page is None else page, boxes[good]).cpu().tolist()
        except (MmeError, OSError, ValueError) as e:
            logger.error(f"Error in batch processing: {e}")  # embedder.py:223-224: every region of the page fails
            return 0
        embedded_count = 0
        for i in range(0, len(good), config.REGION_BATCH_SIZE):  # same upsert granularity as :124-152
            sel = good[i : i + config.REGION_BATCH_SIZE]
            batch_meta = [metas[k] for k in sel]
            documents = [f"Region: {m['region_type']} from {m['parent_image_name']}" for m in batch_meta]
            try:
                self.collection.upsert(ids=[ids[k] for k in sel], embeddings=emb[i : i + len(sel)], documents=documents,
                                       metadatas=batch_meta)
                embedded_count += len(sel)
                logger.info(f"Embedded {len(sel)} regions from {image_filename}")
            except Exception as e:  # :153-154
                logger.error(f"DB Error: {e}")
        return embedded_count
