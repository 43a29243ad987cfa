"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the step in front of the embed hot loop: cutting a detected box out of its page
(deprecated_package/doclayout_detector.py:165-194) and the row region_processor.py:75-113 stores
for it.  Pinned by tests/golden/region_rows.json, which tests/golden/make_golden.py produced by
running the reference's own RegionProcessor.process_image_regions and get_region_image.
Also the class-aware non-maximum suppression that merges the detector's grid passes into the boxes
of a page (3_combine_grids.py:44-137), pinned by tests/golden/nms_cases.json: what the reference's
own apply_non_max_suppression kept on the bundled pages' boxes and on seeded box sets.
"""
from __future__ import annotations

import os

import numpy as np

from .compare import REGION_TYPES_TO_PROCESS


def int_box(box):
    """doclayout_detector.py:179 / region_processor.py:88: `map(int, box)` truncates toward zero."""
    return tuple(int(v) for v in box)


def crop_region(page: np.ndarray, box) -> np.ndarray:
    """`Image.open(page).crop((x0, y0, x1, y1))` (doclayout_detector.py:178-189) on a decoded
    uint8[H, W, 3] page: the crop has the box's size; pixels outside the page are 0."""
    x0, y0, x1, y1 = int_box(box)
    H, W = page.shape[:2]
    out = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0), 3), dtype=np.uint8)
    ys0, ys1, xs0, xs1 = max(y0, 0), min(y1, H), max(x0, 0), min(x1, W)
    if ys1 > ys0 and xs1 > xs0:
        out[ys0 - y0 : ys1 - y0, xs0 - x0 : xs1 - x0] = page[ys0:ys1, xs0:xs1]
    return out


def region_rows(image_path: str, regions: dict):
    """(ids, metadatas, documents) exactly as region_processor.py:75-113,141 builds them."""
    name = os.path.basename(image_path)
    size = regions.get("image_size", {"width": 0, "height": 0})
    ids, metas, docs = [], [], []
    for i, (box, cid, cname, score) in enumerate(zip(regions["boxes"], regions["classes"], regions["class_names"], regions["scores"])):
        if cname not in REGION_TYPES_TO_PROCESS:
            continue
        x0, y0, x1, y1 = int_box(box)
        total = size["width"] * size["height"]
        ids.append(f"region_{os.path.splitext(name)[0]}_{i}")
        metas.append({
            "parent_image": image_path, "parent_image_name": name, "region_index": i, "region_type": cname,
            "region_class_id": int(cid), "region_score": float(score), "box": ",".join(map(str, box)),
            "box_normalized": ",".join(map(str, [x0 / size["width"], y0 / size["height"], x1 / size["width"], y1 / size["height"]])),
            "area_percentage": ((x1 - x0) * (y1 - y0) / total) * 100 if total else 0,
            "width": x1 - x0, "height": y1 - y0, "is_region": True,
        })
        docs.append(f"Region: {cname} from {name}")
    return ids, metas, docs


# ---- the step before the region cache: merging the detector's grid passes (3_combine_grids.py) ----------------------


def box_iou(current: np.ndarray, others: np.ndarray) -> np.ndarray:
    """`calculate_iou(current_box, box)` of 3_combine_grids.py:44-78 for one box against many, in the
    same float64 operation order (`box1` is the box just selected): no overlap -> 0.0 when the
    right edge is LEFT of the left edge (touching boxes take the area branch and get 0 / union);
    union = (area1 + area2) - intersection; union <= 0 -> 0.0."""
    cur = np.asarray(current, dtype=np.float64)
    oth = np.asarray(others, dtype=np.float64).reshape(-1, 4)
    xl = np.maximum(cur[0], oth[:, 0])
    yt = np.maximum(cur[1], oth[:, 1])
    xr = np.minimum(cur[2], oth[:, 2])
    yb = np.minimum(cur[3], oth[:, 3])
    inter = (xr - xl) * (yb - yt)
    a1 = (cur[2] - cur[0]) * (cur[3] - cur[1])
    a2 = (oth[:, 2] - oth[:, 0]) * (oth[:, 3] - oth[:, 1])
    union = (a1 + a2) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(union > 0, inter / np.where(union > 0, union, 1.0), 0.0)
    return np.where((xr < xl) | (yb < yt), 0.0, iou)


def nms_keep(boxes, scores, classes, iou_threshold: float = 0.5) -> list:
    """Indices `apply_non_max_suppression` (3_combine_grids.py:80-137) keeps, in the order it emits them.

    The reference repeatedly takes `scores.index(max(scores))` of what is left -- the FIRST of equal
    scores -- and drops every remaining box of the same class whose IoU with it exceeds the
    threshold: a greedy pass over the boxes in stable descending-score order."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float64)
    classes = np.asarray(classes)
    n = len(scores)
    order = sorted(range(n), key=lambda i: -scores[i])  # sorted() is stable: equal scores keep list order
    alive = np.ones(n, dtype=bool)
    keep = []
    for i in order:
        if not alive[i]:
            continue
        keep.append(int(i))
        alive[i] = False
        rest = np.flatnonzero(alive)
        if len(rest):
            hit = (box_iou(boxes[i], boxes[rest]) > iou_threshold) & (classes[rest] == classes[i])
            alive[rest[hit]] = False
    return keep
