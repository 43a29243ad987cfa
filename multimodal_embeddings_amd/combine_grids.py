"""Merging the detector's grid passes into the boxes of a page (the step in front of the region cache).

Mirrors `apply_non_max_suppression` and `combine_boxes_for_image` of the reference's
`3_combine_grids.py` (:80-137, :199-292): the boxes of every grid pattern of a page
(2x2, 3x3, ... cells, already in page coordinates) are pooled and a class-aware greedy
non-maximum suppression keeps one detection per object.  The reference runs the O(n^2)
`list.index` / `list.pop` loop per page in Python; here the boxes of MANY pages go to the GPU in one
call (K13 `mme_nms_boxes`, one workgroup per page, float64 in the reference's operation order) and
the kept lists come back in the reference's output order.  Visualisation (`cv2`) is out of scope.
"""
from __future__ import annotations

import json
import logging
import os

import numpy as np

from ._lib import Engine, MmeError

logger = logging.getLogger(__name__)

_engine = None


def _default_engine():
    global _engine
    if _engine is None:
        _engine = Engine(0)  # raises MmeError without a GPU / libmme.so: there is no CPU fallback
    return _engine


def apply_non_max_suppression(boxes, scores, classes, class_names, iou_threshold=0.5, engine=None):
    """Same arguments and 4-tuple result as 3_combine_grids.py:80-137 for ONE page."""
    if not boxes:
        return [], [], [], []
    eng = engine or _default_engine()
    (keep,) = eng.nms_boxes(boxes, scores, classes, [0, len(scores)], iou_threshold)
    return ([boxes[i] for i in keep], [scores[i] for i in keep], [classes[i] for i in keep], [class_names[i] for i in keep])


def collect_boxes(json_paths):
    """The pooling half of combine_boxes_for_image (3_combine_grids.py:216-270): grid-info files
    (`cells[*].regions.boxes_original`), grid-cell files (`boxes_original`) and plain detector files
    (`boxes`).  Returns (boxes, scores, classes, class_names, image_path, image_size)."""
    all_boxes, all_scores, all_classes, all_names = [], [], [], []
    image_path = image_size = None
    for path in json_paths:
        try:
            with open(path) as fh:
                data = json.load(fh)
            if "cells" in data:
                if not image_path and "original_image_path" in data:
                    image_path = data["original_image_path"]
                for cell in data["cells"]:
                    reg = cell.get("regions")
                    if reg and "boxes_original" in reg:
                        all_boxes.extend(reg["boxes_original"])
                        all_scores.extend(reg["scores"])
                        all_classes.extend(reg["classes"])
                        all_names.extend(reg["class_names"])
            elif "boxes" in data:
                if not image_path and "image_path" in data:
                    image_path = data["image_path"]
                if not image_size and "image_size" in data:
                    image_size = data["image_size"]
                all_boxes.extend(data["boxes_original"] if "boxes_original" in data else data["boxes"])
                all_scores.extend(data["scores"])
                all_classes.extend(data["classes"])
                all_names.extend(data["class_names"])
        except Exception as e:  # the reference logs and carries on (:269-270)
            logger.error(f"Error reading {path}: {e}")
    return all_boxes, all_scores, all_classes, all_names, image_path, image_size


def combine_boxes_for_images(pages, iou_threshold=0.5, engine=None):
    """`combine_boxes_for_image` (3_combine_grids.py:199-292) for many pages with ONE GPU call.

    pages: dict base name -> list of JSON paths (what `find_grid_jsons` returns).  Result: dict base
    name -> the reference's `combined_regions` dict, or None where a page has no boxes (:272-274)."""
    pooled = {name: collect_boxes(paths) for name, paths in pages.items()}
    names = [n for n, p in pooled.items() if p[0]]
    out = {n: None for n in pages}
    for n in pages:
        if n not in names:
            logger.warning(f"No boxes found for {n}")
    if not names:
        return out
    offs = np.zeros(len(names) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(pooled[n][1]) for n in names])
    if offs[-1] >= 2**31:
        raise MmeError("combine_boxes_for_images: more than 2^31 boxes in one call")
    boxes = np.concatenate([np.asarray(pooled[n][0], dtype=np.float64).reshape(-1, 4) for n in names])
    scores = np.concatenate([np.asarray(pooled[n][1], dtype=np.float64) for n in names])
    classes = np.concatenate([np.asarray(pooled[n][2], dtype=np.int32) for n in names])
    eng = engine or _default_engine()
    keeps = eng.nms_boxes(boxes, scores, classes, offs.astype(np.int32), iou_threshold)
    for n, keep in zip(names, keeps):
        b, s, c, cn, image_path, image_size = pooled[n]
        out[n] = {
            "image_path": image_path,
            "image_size": image_size,
            "parameters": {"iou_threshold": iou_threshold},
            "boxes": [b[i] for i in keep],
            "classes": [c[i] for i in keep],
            "scores": [s[i] for i in keep],
            "class_names": [cn[i] for i in keep],
            "source_jsons": list(pages[n]),
        }
    return out


def combine_boxes_for_image(image_base_name, json_paths, iou_threshold=0.5, engine=None):
    """One page (the reference's signature)."""
    return combine_boxes_for_images({image_base_name: json_paths}, iou_threshold, engine)[image_base_name]
