"""K13 class-aware NMS (mme_nms_boxes) and the combine_grids mirror against what the reference's own
apply_non_max_suppression (3_combine_grids.py:80-137) kept (tests/golden/nms_cases.json) and against the oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from multimodal_embeddings_amd._lib import Engine

    return Engine(0)


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "nms_cases.json")))["cases"]


def test_nms_keeps_what_the_reference_kept_all_pages_in_one_call(engine, golden_dir):
    cases = _cases(golden_dir)
    for thr_i in range(3):
        sel = [c for c in cases if len(c["runs"]) > thr_i]
        by_thr = {}
        for c in sel:
            by_thr.setdefault(c["runs"][thr_i]["iou_threshold"], []).append((c, c["runs"][thr_i]["keep"]))
        for thr, group in by_thr.items():
            offs = np.zeros(len(group) + 1, dtype=np.int32)
            offs[1:] = np.cumsum([len(c["scores"]) for c, _ in group])
            boxes = np.concatenate([np.asarray(c["boxes"], dtype=np.float64).reshape(-1, 4) for c, _ in group])
            scores = np.concatenate([np.asarray(c["scores"], dtype=np.float64) for c, _ in group])
            classes = np.concatenate([np.asarray(c["classes"], dtype=np.int32) for c, _ in group])
            got = engine.nms_boxes(boxes, scores, classes, offs, thr)
            for (c, want), g in zip(group, got):
                assert g.tolist() == want, (c["name"], thr)


def test_nms_random_pages_match_the_oracle(engine):
    from oracle.regions import nms_keep

    rng = np.random.default_rng(5)
    pages = []
    for n in [0, 1, 3, 33, 64, 65, 255, 256, 257, 700, 2100, 0, 12]:
        c = rng.uniform(0, 800, (n, 2))
        wh = rng.uniform(5, 300, (n, 2))
        boxes = np.round(np.concatenate([c - wh / 2, c + wh / 2], axis=1), 1 if n % 2 else 0)
        pages.append((boxes, np.round(rng.uniform(0, 1, n), 2), rng.integers(0, 3, n).astype(np.int32)))
    offs = np.zeros(len(pages) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(p[1]) for p in pages])
    cat = lambda k, shape: np.concatenate([p[k].reshape(shape) for p in pages])
    for thr in (0.5, 0.05, 0.95):
        got = engine.nms_boxes(cat(0, (-1, 4)), cat(1, (-1,)), cat(2, (-1,)), offs, thr)
        for p, g in zip(pages, got):
            assert g.tolist() == nms_keep(p[0], p[1], p[2], thr), (len(p[1]), thr)


def test_combine_grids_mirror_runs_the_reference_call_shape(engine, golden_dir, tmp_path):
    from multimodal_embeddings_amd import combine_grids as cg

    case = next(c for c in _cases(golden_dir) if c["name"].startswith("seeded mixed"))
    names = [f"class{c}" for c in case["classes"]]
    want = case["runs"][0]["keep"]
    b, s, c, n = cg.apply_non_max_suppression(case["boxes"], case["scores"], case["classes"], names, case["runs"][0]["iou_threshold"], engine=engine)
    assert b == [case["boxes"][i] for i in want] and s == [case["scores"][i] for i in want]
    assert c == [case["classes"][i] for i in want] and n == [names[i] for i in want]
    assert cg.apply_non_max_suppression([], [], [], [], engine=engine) == ([], [], [], [])
    # the three JSON shapes combine_boxes_for_image pools (3_combine_grids.py:216-270), split over two files + one unreadable
    half = len(names) // 2
    std = {"image_path": "/pages/p.png", "image_size": {"width": 2000, "height": 3000}, "boxes": case["boxes"][:half], "scores": case["scores"][:half],
           "classes": case["classes"][:half], "class_names": names[:half]}
    grid = {"original_image_path": "/pages/p.png", "cells": [{"regions": {"boxes_original": case["boxes"][half:], "scores": case["scores"][half:],
            "classes": case["classes"][half:], "class_names": names[half:]}}, {"regions": {}}]}
    p1, p2, p3 = tmp_path / "p.json", tmp_path / "p_grid_2x2.json", tmp_path / "broken.json"
    p1.write_text(json.dumps(std))
    p2.write_text(json.dumps(grid))
    p3.write_text("{not json")
    out = cg.combine_boxes_for_images({"p": [str(p1), str(p2), str(p3)], "empty": [str(p3)]}, case["runs"][0]["iou_threshold"], engine=engine)
    assert out["empty"] is None
    assert out["p"]["boxes"] == [case["boxes"][i] for i in want] and out["p"]["class_names"] == [names[i] for i in want]
    assert out["p"]["image_path"] == "/pages/p.png" and out["p"]["image_size"] == std["image_size"] and out["p"]["parameters"] == {"iou_threshold": 0.5}


def test_nms_rejects_bad_arguments(engine):
    from multimodal_embeddings_amd._lib import MmeError

    with pytest.raises(MmeError):
        engine.nms_boxes(np.zeros((2, 4)), np.zeros(2), np.zeros(2, np.int32), [0, 3], 0.5)
    with pytest.raises(MmeError):
        engine.nms_boxes(np.zeros((2, 4)), np.zeros(2), np.zeros(2, np.int32), [0, 2], float("nan"))
    assert engine.nms_boxes(np.zeros((0, 4)), np.zeros(0), np.zeros(0, np.int32), [0], 0.5) == []
    # a NaN score (json.load accepts it) would leave the descending-score rank without a total order: rejected by name,
    # and the call after it works (the walk used to index with never-written order[] slots)
    rng = np.random.default_rng(5)
    xy = rng.uniform(0, 900, (300, 2))
    boxes = np.concatenate([xy, xy + rng.uniform(5, 120, (300, 2))], axis=1)
    scores = rng.uniform(0.1, 1.0, 300)
    bad = scores.copy()
    bad[[7, 150, 299]] = np.nan
    with pytest.raises(MmeError, match="NaN"):
        engine.nms_boxes(boxes, bad, np.zeros(300, np.int32), [0, 300], 0.5)
    kept = engine.nms_boxes(boxes, scores, np.zeros(300, np.int32), [0, 300], 0.5)
    from oracle import regions as oreg

    assert kept[0].tolist() == oreg.nms_keep(boxes.tolist(), scores.tolist(), [0] * 300, 0.5)
    inf = scores.copy()
    inf[3], inf[4] = np.inf, -np.inf  # infinities are ordered: first and last of the walk
    kept = engine.nms_boxes(boxes, inf, np.zeros(300, np.int32), [0, 300], 0.5)
    assert kept[0][0] == 3 and kept[0].tolist() == oreg.nms_keep(boxes.tolist(), inf.tolist(), [0] * 300, 0.5)
