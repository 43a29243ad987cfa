"""Tile-ViT encoder option: the Mllama vision tower geometry of the reference's checkpoint on the GPU
(mme_load_tile_vit / mme_tile_vit_forward) against the CPU oracle (oracle/mllama_vision.py, itself pinned to
transformers' MllamaVisionModel) and against rows recorded from that class itself (tests/golden/tile_vit_cases.npz)."""
import json
import os
from dataclasses import replace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodal_embeddings_amd.weights import TILE_VIT, make_tile_vit_weights  # noqa: E402


def _image(source):
    seed, hw = source.split(":")[1:]
    h, w = map(int, hw.split("x"))
    return np.random.default_rng(int(seed)).integers(0, 256, (h, w, 3), dtype=np.uint8)


def _token_cos(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)


def _prep(engine, arrays):
    from multimodal_embeddings_amd.embedder import RegionEmbedder

    emb = RegionEmbedder(engine=engine)
    pix, offs, hw = emb.pack(arrays)
    return engine.preprocess_tiles(pix, offs, hw, 560, 4)


def test_shallow_tower_matches_the_oracle_on_padded_and_full_images():
    """2 local + 2 gated global layers, intermediate states after local layers 0 and 1: every kernel of the option
    (patchify, assembly with gated embeddings, LayerNorm-folded GEMMs with d = 1280 statistics, flash attention over
    6432 tokens with the (padding, padding) mask, layernorm_post, gated branches, output gather, pooling) against the
    fp32 CPU oracle on a 1-tile, a 2-tile and a 4-tile image.  Tolerance: 1e-3 cosine per token (north_star)."""
    from multimodal_embeddings_amd._lib import Engine
    from oracle import mllama_vision as om

    geom = replace(TILE_VIT, num_layers=2, num_global_layers=2, intermediate_layers=(0, 1))
    w = make_tile_vit_weights(3, geom)
    eng = Engine(0)
    eng.load_tile_vit(w, geom)
    rng = np.random.default_rng(4)
    arrays = [rng.integers(0, 256, s, dtype=np.uint8) for s in [(300, 200, 3), (400, 900, 3), (1000, 1100, 3)]]
    pv, ids, mask, nt = _prep(eng, arrays)
    assert nt == [1, 2, 4]
    for mode in (2, 1):  # LayerNorm statistics from the producing GEMM's partial sums / from a pass over x: same bits
        eng.set_ln_fusion(mode)
        hidden, e32, e16 = eng.tile_vit_forward(pv, ids, nt, want_hidden=True)
        torch.cuda.synchronize()
        if mode == 2:
            first = hidden.clone()
        else:
            assert torch.equal(first, hidden)
    host = hidden.cpu().numpy()
    assert host.shape == (3, 4, 1601, geom.output_dim) and np.isfinite(host).all()
    pvh = pv.cpu().numpy()
    for k in range(3):
        want = om.vision_forward(pvh[k], int(ids[k]), nt[k], w, geom)
        cos = _token_cos(host[k, : nt[k]], want[: nt[k]])
        assert cos.min() >= 1 - 1e-3, (k, float(cos.min()))
        # the padding tiles come out of the model too (they are keys and values of every layer): same bar
        if nt[k] < 4:
            assert _token_cos(host[k, nt[k] :], want[nt[k] :]).min() >= 1 - 1e-3
        pooled = om.pooled_embedding(want)
        assert 1.0 - float(np.dot(e32[k].cpu().numpy().astype(np.float64), pooled)) <= 1e-3
        assert abs(float(np.linalg.norm(e32[k].cpu().numpy())) - 1.0) <= 1e-5
    assert torch.equal(e16.float(), e32.to(torch.bfloat16).float())
    eng.close()


def test_intermediate_save_point_conventions_match_the_oracle():
    """`intermediate[k] = i` may name the state AFTER local layer i (transformers 5.15, the pinned convention)
    or the state ENTERING it (encoders that record before running a layer).  Both are explicit in
    mme_tile_vit_weights.intermediate_save_point; each is checked against the oracle's restatement of the same
    convention (the "before" form is unpinned to any transformers release here: parity unpinned, said in mme.h), and
    the two differ exactly in the intermediate features while the final-state features are identical."""
    from multimodal_embeddings_amd._lib import Engine, MmeError
    from oracle import mllama_vision as om

    rng = np.random.default_rng(8)
    arrays = [rng.integers(0, 256, (500, 640, 3), dtype=np.uint8)]
    out = {}
    for point in ("after", "before"):
        geom = replace(TILE_VIT, num_layers=3, num_global_layers=1, intermediate_layers=(0, 2), intermediate_save_point=point)
        w = make_tile_vit_weights(5, geom)
        eng = Engine(0)
        eng.load_tile_vit(w, geom)
        pv, ids, mask, nt = _prep(eng, arrays)
        hidden, e32, _ = eng.tile_vit_forward(pv, ids, nt, want_hidden=True)
        torch.cuda.synchronize()
        host = hidden.cpu().numpy()[0]
        want = om.vision_forward(pv.cpu().numpy()[0], int(ids[0]), nt[0], w, geom)
        assert _token_cos(host[: nt[0]], want[: nt[0]]).min() >= 1 - 1e-3, point
        out[point] = host
        eng.close()
    D = TILE_VIT.hidden_size
    a, b = out["after"][:, :, :D], out["before"][:, :, :D]
    assert np.array_equal(a, b)  # the final state does not depend on the convention
    ia = out["after"][:, :, D:].reshape(4, 1601, D, 2)
    ib = out["before"][:, :, D:].reshape(4, 1601, D, 2)
    assert not np.array_equal(ia[..., 0], ib[..., 0])
    # "before layer 2" is "after layer 1"; a third run saving after layers (1,) must reproduce it bit for bit
    geom = replace(TILE_VIT, num_layers=3, num_global_layers=1, intermediate_layers=(1,), intermediate_save_point="after")
    eng = Engine(0)
    eng.load_tile_vit(make_tile_vit_weights(5, geom), geom)
    pv, ids, mask, nt = _prep(eng, arrays)
    hidden, _, _ = eng.tile_vit_forward(pv, ids, nt, want_hidden=True)
    torch.cuda.synchronize()
    assert np.array_equal(hidden.cpu().numpy()[0][:, :, D:], ib[..., 1])
    eng.close()
    with pytest.raises((MmeError, KeyError)):
        bad = replace(TILE_VIT, num_layers=1, num_global_layers=0, intermediate_layers=(0,), intermediate_save_point="sideways")
        Engine(0).load_tile_vit(make_tile_vit_weights(5, bad), bad)


def test_peaked_attention_moves_the_running_maximum_mid_sequence():
    """The attention kernel computes a granule's probabilities against the reference point the query already has and only
    moves it (rescale + second pass) when a score lands more than 2^8 above it.  With the seeded weights scores spread
    over ~3 log2 units, so that path runs once per query; here q_proj and k_proj are scaled by 4 (scores x 16: tens of
    log2 units of spread), which moves the maximum again and again along the 6432 keys -- on a 4-tile image and on a
    2-tile image whose padding tiles take the masked form.  Same bar as the other tests: 1e-3 cosine per token."""
    from multimodal_embeddings_amd._lib import Engine
    from oracle import mllama_vision as om

    geom = replace(TILE_VIT, num_layers=1, num_global_layers=1, intermediate_layers=(0,))
    w = dict(make_tile_vit_weights(5, geom))
    for name in list(w):
        if name.endswith("self_attn.q_proj.weight") or name.endswith("self_attn.k_proj.weight"):
            w[name] = (w[name] * np.float32(4.0)).astype(np.float32)  # a power of two: still bf16-representable
    eng = Engine(0)
    eng.load_tile_vit(w, geom)
    rng = np.random.default_rng(9)
    arrays = [rng.integers(0, 256, s, dtype=np.uint8) for s in [(1000, 1100, 3), (400, 900, 3)]]
    pv, ids, mask, nt = _prep(eng, arrays)
    assert nt == [4, 2]
    hidden, e32, _ = eng.tile_vit_forward(pv, ids, nt, want_hidden=True)
    torch.cuda.synchronize()
    host = hidden.cpu().numpy()
    assert np.isfinite(host).all()
    pvh = pv.cpu().numpy()
    for k in range(2):
        want = om.vision_forward(pvh[k], int(ids[k]), nt[k], w, geom)
        cos = _token_cos(host[k], want)
        assert cos.min() >= 1 - 1e-3, (k, float(cos.min()))
    eng.close()


def test_fast_attention_form_tracks_its_range_and_falls_back_to_the_exact_kernel():
    """The default attention launch of the option is the FAST form (attention_tiles.hip: reference point inside the matrix
    pipe, row sums from a ones column beside V, the reference re-centred from the row sum instead of a running maximum)
    followed by the exact kernel, which runs only when a row of the fast launch left its range (guard word per layer).
      * ordinary weights: fast == exact to 1e-4 cosine per token (both are within 1e-3 of the oracle, test above);
      * q_proj / k_proj scaled by 4 ... 16 (scores x 16 ... x 256: first the row sums pass 2^60 and the reference is
        re-centred, then single scores jump past what a tile can absorb and the guard fires): every scale agrees with
        the exact kernel to 1e-4, and once the guard has fired the output IS the exact kernel's, bit for bit;
      * mode 2 forces the guard: bit-identical to the exact kernel at ordinary weights.
    A 4-tile image and a 2-tile image (padding tiles: the masked form of a sub-tile)."""
    from multimodal_embeddings_amd._lib import Engine

    geom = replace(TILE_VIT, num_layers=1, num_global_layers=1, intermediate_layers=(0,))
    base = dict(make_tile_vit_weights(5, geom))
    rng = np.random.default_rng(9)
    arrays = [rng.integers(0, 256, s, dtype=np.uint8) for s in [(1000, 1100, 3), (400, 900, 3)]]
    bitwise = []
    for scale in (1.0, 4.0, 6.0, 8.0, 12.0, 16.0):
        w = dict(base)
        for name in list(w):
            if name.endswith("self_attn.q_proj.weight") or name.endswith("self_attn.k_proj.weight"):
                w[name] = (w[name] * np.float32(scale)).astype(np.float32)
        eng = Engine(0)
        eng.load_tile_vit(w, geom)
        pv, ids, mask, nt = _prep(eng, arrays)
        out = {}
        for mode in ((0, 1, 2) if scale == 1.0 else (0, 1)):
            eng.set_attention_mode(mode)
            hidden, _, _ = eng.tile_vit_forward(pv, ids, nt, want_hidden=True)
            torch.cuda.synchronize()
            out[mode] = hidden.cpu().numpy()
            assert np.isfinite(out[mode]).all(), (scale, mode)
        cos = _token_cos(out[1].reshape(-1, out[1].shape[-1]), out[0].reshape(-1, out[0].shape[-1]))
        assert cos.min() >= 1 - 1e-4, (scale, float(cos.min()))
        bitwise.append(bool(np.array_equal(out[0], out[1])))
        if scale == 1.0:
            assert np.array_equal(out[2], out[0])  # forced guard: the exact kernel's output stands
            assert not bitwise[-1]  # ... and the fast form really is another kernel
        eng.close()
    assert bitwise[-1], bitwise  # scores x 256: the guard fired in every layer, the exact kernel's output stands
    print("fast form bit-identical to the exact kernel per q/k scale (1, 4, 6, 8, 12, 16):", bitwise)


def test_full_tower_matches_rows_recorded_from_transformers(golden_dir):
    """The full 32 + 8 layer tower with the seeded weights of make_tile_vit_weights(2) on one image per tile
    arrangement (all eight aspect-ratio ids): rows of tokens {0, 1, 800, 1600} of every real tile against what
    transformers' MllamaVisionModel(MllamaVisionConfig(image_size=560)) produced for the same pixels (fp32, CPU; recorded
    by tests/golden/make_golden.py --only-tile-vit).  1e-3 cosine per token, bf16 arithmetic through 40 layers."""
    from multimodal_embeddings_amd._lib import Engine

    path = os.path.join(golden_dir, "tile_vit_cases.npz")
    g = np.load(path)
    sources = [str(s) for s in g["sources"]]
    tokens = g["tokens"].tolist()
    cases = json.load(open(os.path.join(golden_dir, "tile_cases.json")))["cases"]
    by_source = {c["source"]: c for c in cases}
    eng = Engine(0)
    eng.load_tile_vit(make_tile_vit_weights(2))
    eng.set_chunk(4)
    arrays = [_image(s) for s in sources]
    pv, ids, mask, nt = _prep(eng, arrays)
    assert [int(i) for i in ids] == [by_source[s]["aspect_ratio_id"] for s in sources] == list(range(1, 9))
    hidden, e32, _ = eng.tile_vit_forward(pv, ids, nt, want_hidden=True)  # two passes of four images
    torch.cuda.synchronize()
    worst = 1.0
    for k, s in enumerate(sources):
        want = g[f"rows_{by_source[s]['aspect_ratio_id']}"].astype(np.float32)  # [tiles, 4, 7680]
        got = hidden[k, : nt[k]][:, tokens].cpu().numpy()
        cos = _token_cos(got, want)
        worst = min(worst, float(cos.min()))
        assert cos.min() >= 1 - 1e-3, (s, float(cos.min()))
        cls = want[0, 0].astype(np.float64)
        assert 1.0 - float(np.dot(e32[k].cpu().numpy().astype(np.float64), cls / np.linalg.norm(cls))) <= 1e-3
    print("tile-ViT full tower: worst token cosine", worst)
    eng.close()


def test_bench_config_tilevit_prints_the_contract_line():
    """`bench.py --config tilevit` (a fresh child process): one JSON line with the contract's keys, `roofline` for the attention
    kernel and `roofline_gemm`, both with live HIP-event durations."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "tilevit", "--crops", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["config"]["name"] == "tilevit" and d["unit"] == "region-crops/s" and d["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and 0.05 < d["roofline"]["frac"] < 1.0 and d["roofline"]["avg_launch_ms"] > 0
    assert 0.05 < d["roofline_gemm"]["frac"] < 1.0
