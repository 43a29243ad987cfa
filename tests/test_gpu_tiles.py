"""Mllama multi-tile preprocessing on the GPU (mme_preprocess_tiles) against transformers' own output
(tests/golden/tile_cases.json: sha256 of every f32 pixel value) and the oracle restatement."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def embedder():
    from multimodal_embeddings_amd.embedder import RegionEmbedder

    return RegionEmbedder()


def _images(golden_dir, g):
    from PIL import Image

    for c in g["cases"]:
        kind, rest = c["source"].split(":", 1)
        if kind == "file":
            yield c, np.array(Image.open(os.path.join(golden_dir, "crops", rest)).convert("RGB"))
        else:
            seed, hw = rest.split(":")
            h, w = map(int, hw.split("x"))
            yield c, np.random.default_rng(int(seed)).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_tiles_bit_exact_against_transformers_golden(embedder, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "tile_cases.json")))
    cases, arrays = zip(*_images(golden_dir, g))
    probe = np.array(g["probe_index"])
    # in two batches so that canvases of different widths share a launch
    for lo in range(0, len(arrays), 23):
        batch = list(arrays[lo : lo + 23])
        pix, offs, hw = embedder.pack(batch)
        pv, ids, mask, nt = embedder.engine.preprocess_tiles(pix, offs, hw, g["tile"], g["max_tiles"])
        torch.cuda.synchronize()
        host = pv.cpu().numpy()
        for k, c in enumerate(cases[lo : lo + 23]):
            assert int(ids[k]) == c["aspect_ratio_id"] and nt[k] == c["num_tiles"] and mask[k].tolist() == c["aspect_ratio_mask"], c["source"]
            assert host[k].reshape(-1)[probe].tolist() == c["probe"], c["source"]
            assert hashlib.sha256(np.ascontiguousarray(host[k]).tobytes()).hexdigest() == c["sha256"], c["source"]


def test_tiles_other_geometries_match_oracle(embedder):
    """tile 224 x 1 tile reproduces the single-tile canvas K1 patchifies; tile 336 x 6 tiles and 16 x 16 exercise
    other grids.  Bit-exact against the oracle restatement."""
    from oracle import preprocess as opre

    rng = np.random.default_rng(12)
    arrays = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(300, 170), (17, 900), (224, 224), (1000, 999), (40, 33)]]
    pix, offs, hw = embedder.pack(arrays)
    for tile, mt in [(224, 1), (336, 6), (16, 16)]:
        pv, ids, mask, nt = embedder.engine.preprocess_tiles(pix, offs, hw, tile, mt)
        host = pv.cpu().numpy()
        for k, a in enumerate(arrays):
            want, aid, n_t, _ = opre.preprocess_tiles(a, tile, mt)
            assert int(ids[k]) == aid and nt[k] == n_t and np.array_equal(host[k], want), (tile, mt, k)
    single = embedder.engine.preprocess_tiles(pix, offs, hw, 224, 1)[0].cpu().numpy()
    assert np.array_equal(single[0, 0], opre.preprocess_crop(arrays[0]))


def test_tiles_rejects_bad_arguments(embedder):
    from multimodal_embeddings_amd._lib import MmeError

    pix, offs, hw = embedder.pack([np.zeros((10, 10, 3), np.uint8)])
    for tile, mt in [(0, 4), (561, 4), (560, 0), (560, 17)]:
        with pytest.raises(MmeError):
            embedder.engine.preprocess_tiles(pix, offs, hw, tile, mt)


def test_process_images_has_the_processor_output_shapes(embedder, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "tile_cases.json")))
    cases, arrays = zip(*list(_images(golden_dir, g))[:5])
    out = embedder.process_images(list(arrays))
    assert tuple(out["pixel_values"].shape) == (5, 1, 4, 3, 560, 560) and out["pixel_values"].dtype == torch.float32
    assert out["aspect_ratio_ids"].shape == (5, 1) and out["aspect_ratio_mask"].shape == (5, 1, 4)
    assert out["aspect_ratio_ids"][:, 0].tolist() == [c["aspect_ratio_id"] for c in cases]
    assert out["num_tiles"] == [[c["num_tiles"]] for c in cases]
