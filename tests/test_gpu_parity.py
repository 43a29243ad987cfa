"""GPU parity tests: HIP path (through the C ABI) vs the oracle and the golden fixtures.

Run on the MI355X box with `pytest -m gpu`.  Tolerances (BASELINE.json north_star):
  * uint8 resize / patch matrix: bit-exact (the bf16 patch values are exact roundings of
    the oracle's f32 pixel_values);
  * embeddings: 1 - cos(gpu, oracle fp32) <= 1e-3 (bf16 MFMA path vs fp32 CPU oracle);
    a stricter centred check guards against the trivial pass that near-identical
    random-weight embeddings would allow;
  * cosine matrix: |gpu - f64| <= 2e-6 (bf16 inputs are exact, f32 accumulate);
  * page matrix: 1e-9 absolute on the normalised S given identical bf16 embeddings.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodal_embeddings_amd.weights import bf16_bits_to_f32, make_vit_weights, round_to_bf16, synthetic_crops  # noqa: E402


@pytest.fixture(scope="module")
def engine():
    from multimodal_embeddings_amd._lib import Engine

    e = Engine(0)
    e.load_vit(make_vit_weights(seed=1))
    yield e
    e.close()


@pytest.fixture(scope="module")
def engine_hot():
    """Weights with 4x the init std: peaky softmax, large GELU arguments."""
    from multimodal_embeddings_amd._lib import Engine

    e = Engine(0)
    e.load_vit(make_vit_weights(seed=1, std=0.08))
    yield e
    e.close()


def _pack(arrays, device="cuda:0"):
    hw = np.array([a.shape[:2] for a in arrays], dtype=np.int32).reshape(-1, 2)
    sizes = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
    offs = np.zeros(len(arrays), dtype=np.int64)
    offs[1:] = np.cumsum((sizes[:-1] + 15) // 16 * 16)
    buf = np.zeros(int(offs[-1] + sizes[-1]) + 16, dtype=np.uint8)
    for a, o, s in zip(arrays, offs, sizes):
        buf[o : o + s] = a.reshape(-1)
    return torch.from_numpy(buf).to(device), offs, hw


def _bf16_tensor_to_f32(t):
    return t.float().cpu().numpy()


def _golden_crops(golden_dir):
    from PIL import Image

    man = json.load(open(os.path.join(golden_dir, "crops_manifest.json")))
    return [np.array(Image.open(os.path.join(golden_dir, "crops", c["file"])).convert("RGB")) for c in man["crops"]], man


def test_preprocess_bit_exact_real_crops(engine, golden_dir):
    from oracle import preprocess as opre

    arrays, man = _golden_crops(golden_dir)
    pix, offs, hw = _pack(arrays)
    patches = engine.preprocess(pix, offs, hw)
    torch.cuda.synchronize()
    got = _bf16_tensor_to_f32(patches).reshape(len(arrays), 196, 768)
    for i, a in enumerate(arrays):
        want = round_to_bf16(opre.preprocess_to_patches(a))
        assert np.array_equal(got[i], want), man["crops"][i]["file"]


def test_preprocess_bit_exact_synthetic_shapes(engine):
    from oracle import preprocess as opre

    rng = np.random.default_rng(11)
    shapes = [(224, 224), (20, 63), (63, 20), (1, 1), (1, 300), (300, 1), (5114, 60), (37, 3862), (223, 225), (225, 223),
              (448, 448), (100, 100), (1000, 333), (224, 100), (100, 224), (17, 8000), (8000, 9), (2000, 1999), (16, 16), (500, 224),
              # one to three output rows / columns of an extreme aspect ratio: up to 2 * 72 + 1 taps per output coordinate
              (71, 8000), (106, 8000), (140, 8000), (8000, 106), (8000, 140), (1400, 1350), (1351, 700), (3000, 2049)]
    arrays = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    pix, offs, hw = _pack(arrays)
    patches = engine.preprocess(pix, offs, hw)
    torch.cuda.synchronize()
    got = _bf16_tensor_to_f32(patches).reshape(len(arrays), 196, 768)
    for i, a in enumerate(arrays):
        want = round_to_bf16(opre.preprocess_to_patches(a))
        assert np.array_equal(got[i], want), shapes[i]


def test_preprocess_rejects_oversize(engine):
    from multimodal_embeddings_amd._lib import MmeError

    pix = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(MmeError):
        engine.preprocess(pix, np.array([0]), np.array([[8001, 1]]))
    with pytest.raises(MmeError):
        engine.preprocess(pix, np.array([0]), np.array([[0, 5]]))


def _oracle_embed(arrays, w, pool):
    from oracle import preprocess as opre
    from oracle import vit as ovit

    patches = np.stack([opre.preprocess_to_patches(a) for a in arrays])
    return ovit.vit_embed(patches, w, pool=pool)


def _check_embeddings(got, want, tol=1e-3):
    got = got / np.linalg.norm(got, axis=1, keepdims=True)
    cos = np.sum(got * want, axis=1)
    assert np.all(1.0 - cos <= tol), (1.0 - cos).max()
    # centred check: remove the common component so near-identical embeddings cannot pass trivially
    if len(got) >= 4:
        mu = want.mean(axis=0, keepdims=True)
        g, w_ = got - mu, want - mu
        ccos = np.sum(g * w_, axis=1) / (np.linalg.norm(g, axis=1) * np.linalg.norm(w_, axis=1))
        assert np.all(ccos > 0.98), ccos.min()
    return float((1.0 - cos).max())


@pytest.mark.parametrize("pool,token", [("cls", 0), ("last", 196)])
def test_embed_real_crops_within_1e3_cosine(engine, golden_dir, pool, token):
    arrays, _ = _golden_crops(golden_dir)
    pix, offs, hw = _pack(arrays)
    e32, e16 = engine.embed(pix, offs, hw, pool_token=token)
    torch.cuda.synchronize()
    want = _oracle_embed(arrays, make_vit_weights(seed=1), pool)
    _check_embeddings(e32.cpu().numpy(), want)
    # bf16 copy is the rounding of the f32 copy, and rows are unit length
    assert np.array_equal(_bf16_tensor_to_f32(e16), round_to_bf16(e32.cpu().numpy()))
    assert np.allclose(np.linalg.norm(e32.cpu().numpy(), axis=1), 1.0, atol=1e-5)


def test_ln_fusion_on_off_agree(engine, golden_dir):
    """LayerNorm folded into the GEMMs vs the separate LayerNorm kernel: same embeddings up to bf16 noise."""
    arrays, _ = _golden_crops(golden_dir)
    pix, offs, hw = _pack(arrays)
    want = _oracle_embed(arrays, make_vit_weights(seed=1), "cls")
    engine.set_ln_fusion(False)
    a, _ = engine.embed(pix, offs, hw)
    engine.set_ln_fusion(True)
    b, _ = engine.embed(pix, offs, hw)
    torch.cuda.synchronize()
    ea = _check_embeddings(a.cpu().numpy(), want)
    eb = _check_embeddings(b.cpu().numpy(), want)
    assert float((1.0 - (a * b).sum(dim=1)).max()) <= 2e-4, (ea, eb)
    # folded LayerNorm: statistics from one pass over x (mode 1) and from the partial sums the producing GEMM's
    # epilogue leaves (mode 2) follow ONE canonical summation order -> bit-identical embeddings, also on a batch
    # with a ragged last row tile (300 crops = 59 100 rows = 230 tiles of 256 + 220 rows) and for every GEMM variant
    crops = synthetic_crops(300, seed=9)
    pixb, offsb, hwb = _pack(list(crops))
    engine.set_ln_fusion(1)
    ref, _ = engine.embed(pixb, offsb, hwb)
    for variant in (0, 1, 3, 4):
        engine.set_gemm_variant(variant)
        for mode in (2, 1):
            engine.set_ln_fusion(mode)
            got, _ = engine.embed(pixb, offsb, hwb)
            assert torch.equal(ref, got), (variant, mode)
    engine.set_gemm_variant(0)
    engine.set_ln_fusion(2)
    want300 = _oracle_embed(list(crops[:6]), make_vit_weights(seed=1), "cls")
    _check_embeddings(ref[:6].cpu().numpy(), want300)


def test_embed_hot_weights(engine_hot, golden_dir):
    arrays, _ = _golden_crops(golden_dir)
    arrays = arrays[:12]
    pix, offs, hw = _pack(arrays)
    e32, _ = engine_hot.embed(pix, offs, hw)
    torch.cuda.synchronize()
    want = _oracle_embed(arrays, make_vit_weights(seed=1, std=0.08), "cls")
    _check_embeddings(e32.cpu().numpy(), want)


def test_attention_fast_form_guard_and_exact_rerun(engine, golden_dir):
    """K5's default form takes the exponentials of a query row against the maximum over its FIRST 32 keys (softmax is
    invariant to the reference point); a row whose sum leaves [1, 2^100) raises a guard word and the launch is redone
    by the exact kernel.  (1) fast and exact agree to rounding and both meet the oracle; no layer is redone on ordinary
    inputs; (2) with the guard forced (mode 2) every layer is redone and the result is the exact kernel's bit for bit;
    (3) query / key weights scaled so that scores spread over hundreds of log2 units trip the guard for real: finite
    outputs, the redone layers flagged, and the result is the exact kernel's (bit for bit when every layer was redone,
    to rounding otherwise; no oracle bar here: at such scores the bf16 rounding of Q and K alone moves an arg-max)."""
    from multimodal_embeddings_amd._lib import Engine

    arrays, _ = _golden_crops(golden_dir)
    arrays = arrays[:12]
    pix, offs, hw = _pack(arrays)
    w = make_vit_weights(seed=1)
    want = _oracle_embed(arrays, w, "cls")
    engine.set_attention_mode("exact")
    exact, _ = engine.embed(pix, offs, hw)
    assert engine.attention_redone() == [0] * 12
    engine.set_attention_mode("fast")
    fast, _ = engine.embed(pix, offs, hw)
    assert engine.attention_redone() == [0] * 12
    engine.set_attention_mode("fast_forced_redo")
    forced, _ = engine.embed(pix, offs, hw)
    assert engine.attention_redone() == [1] * 12
    engine.set_attention_mode("fast")
    _check_embeddings(exact.cpu().numpy(), want)
    _check_embeddings(fast.cpu().numpy(), want)
    assert torch.equal(forced, exact)
    assert float((1.0 - (fast * exact).sum(dim=1)).max()) <= 1e-4
    # (3) scores far outside the range a tile-0 reference point covers
    wild = {k: (v * 16.0 if (".q_proj.weight" in k or ".k_proj.weight" in k) else v) for k, v in w.items()}
    assert any(not np.array_equal(wild[k], w[k]) for k in w), sorted(w)[:8]
    e = Engine(0)
    e.load_vit(wild)
    got, _ = e.embed(pix, offs, hw)
    redone = e.attention_redone()
    e.set_attention_mode("exact")
    ref, _ = e.embed(pix, offs, hw)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and torch.isfinite(ref).all() and sum(redone) >= 1, redone
    assert float((1.0 - (got * ref).sum(dim=1)).max()) <= 1e-3, redone
    if all(redone):
        assert torch.equal(got, ref)
    e.close()


def test_tile_order_modes_are_bit_identical(engine):
    """mme_set_tile_order changes only the ORDER in which the GEMMs walk their row panels and the attention its crops."""
    crops = synthetic_crops(300, seed=21)
    pix, offs, hw = _pack(list(crops))
    try:
        outs = []
        for mode in (1, 0, 2):
            engine.set_tile_order(mode)
            outs.append(engine.embed(pix, offs, hw)[0].clone())
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    finally:
        engine.set_tile_order(1)


def test_last_layer_pruning_is_bit_identical(engine):
    """mme_set_forward_pruning: after the last layer's attention only the pooled token's row is ever read (K8), so only the
    query block that holds it is attended and o_proj / LayerNorm / fc1 / fc2 of that layer run on the n gathered rows.  Same
    kernels, same per-row arithmetic -> embeddings bit-identical to the full pass, for both pooling rules, both LayerNorm
    statistics modes, both softmax forms, a ragged batch and a single crop."""
    crops = synthetic_crops(300, seed=13)
    pix, offs, hw = _pack(list(crops))
    try:
        for attn in ("fast", "exact"):
            engine.set_attention_mode(attn)
            for mode in (2, 1):
                engine.set_ln_fusion(mode)
                for tok in (0, 196, 77):
                    engine.set_forward_pruning(False)
                    full32, full16 = engine.embed(pix, offs, hw, tok)
                    engine.set_forward_pruning(True)
                    got32, got16 = engine.embed(pix, offs, hw, tok)
                    torch.cuda.synchronize()
                    assert torch.equal(full32, got32) and torch.equal(full16.view(torch.int16), got16.view(torch.int16)), (attn, mode, tok)
        engine.set_forward_pruning(False)
        one_full, _ = engine.embed(pix[: 150528 + 16], offs[:1], hw[:1])
        engine.set_forward_pruning(True)
        one, _ = engine.embed(pix[: 150528 + 16], offs[:1], hw[:1])
        assert torch.equal(one, one_full)
    finally:
        engine.set_forward_pruning(False)
        engine.set_attention_mode("fast")
        engine.set_ln_fusion(2)


def test_embed_matches_transformers_golden(engine, golden_dir):
    """vit_cases.npz was produced by transformers.ViTModel itself (make_golden.py)."""
    g = np.load(os.path.join(golden_dir, "vit_cases.npz"))
    crops = synthetic_crops(int(g["n"]), seed=int(g["crop_seed"]))
    pix, offs, hw = _pack(list(crops))
    for pool, token in (("cls", 0), ("last", 196)):
        e32, _ = engine.embed(pix, offs, hw, pool_token=token)
        torch.cuda.synchronize()
        cos = np.sum(e32.cpu().numpy() * g[f"std002_{pool}"], axis=1)
        assert np.all(1.0 - cos <= 1e-3), (pool, 1.0 - cos)


def test_embed_chunking_and_ragged_batch(engine):
    """Results must not depend on the chunk size or on the position inside a batch."""
    crops = synthetic_crops(37, seed=3)
    pix, offs, hw = _pack(list(crops))
    engine.set_chunk(1024)
    a, _ = engine.embed(pix, offs, hw)
    engine.set_chunk(8)
    b, _ = engine.embed(pix, offs, hw)
    engine.set_chunk(1024)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    c, _ = engine.embed(pix, offs[5:6], hw[5:6])
    assert torch.equal(a[5:6], c)
    e0, _ = engine.embed(pix, offs[:0], hw[:0])
    assert e0.shape == (0, 768)


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
def test_cosine_matches_f64(engine, variant):
    engine.set_gemm_variant(variant)
    rng = np.random.default_rng(2)
    for m, n, d in [(1, 1, 64), (5, 300, 768), (257, 129, 768), (1000, 1000, 128), (130, 4100, 768), (1024, 2304, 768), (777, 1028, 128)]:
        a = rng.standard_normal((m, d)).astype(np.float32)
        b = rng.standard_normal((n, d)).astype(np.float32)
        ta = engine.normalise_rows(torch.from_numpy(a).cuda())
        tb = engine.normalise_rows(torch.from_numpy(b).cuda())
        sim = engine.cosine(ta, tb)
        torch.cuda.synchronize()
        fa, fb = ta.float().cpu().numpy().astype(np.float64), tb.float().cpu().numpy().astype(np.float64)
        want = fa @ fb.T
        assert np.abs(sim.cpu().numpy() - want).max() <= 2e-6, (m, n, d)
        if n % 4 == 0:  # the bf16-S option is the round-to-nearest-even of the f32 block, bit for bit (|error| <= 2^-9 relative)
            s16 = engine.cosine_bf16(ta, tb)
            assert torch.equal(s16, sim.to(torch.bfloat16)), (m, n, d)
            assert float((s16.float() - sim).abs().max()) <= 2.0 ** -9
        # normalise_rows itself: unit rows, rounding of the f32 normalisation
        ref = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
        assert np.abs(fa - ref).max() <= 2.0 ** -8
    engine.set_gemm_variant(0)


def test_gemm_variants_bit_identical_and_race_free(engine, golden_dir):
    """128x128 and 256x256 kernels use the same MFMA and K order: outputs must be bit-identical.

    The 256 kernel keeps LDS-DMA loads in flight across barriers (counted vmcnt); a misplaced
    wait shows up as rare wrong tiles, so large shapes are repeated and compared bitwise.
    """
    rng = np.random.default_rng(7)
    for m, n, d, reps in [(4096, 4096, 768, 3), (2048, 1024, 3072, 3), (777, 1300, 64, 2), (5000, 300, 1024, 2)]:
        a = engine.normalise_rows(torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda())
        b = engine.normalise_rows(torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda())
        engine.set_gemm_variant(1)
        ref = engine.cosine(a, b)
        for variant in (2, 3, 4):  # 4: the last row block of a wave tile is stored from inside the next tile's K loop
            engine.set_gemm_variant(variant)
            for _ in range(reps):
                got = engine.cosine(a, b)
                assert torch.equal(got, ref), (variant, m, n, d)
    arrays, _ = _golden_crops(golden_dir)
    pix, offs, hw = _pack(arrays)
    engine.set_gemm_variant(1)
    e1, _ = engine.embed(pix, offs, hw)
    for variant in (2, 3, 4, 5):  # 4 / 5: a quarter / half of each tile's stores deferred into the next tile's K loop
        engine.set_gemm_variant(variant)
        for _ in range(3):
            e2, _ = engine.embed(pix, offs, hw)
            assert torch.equal(e1, e2), variant
    # a batch large enough for several output tiles per workgroup (K-tile stream across tiles)
    crops = synthetic_crops(300, seed=5)
    pixb, offsb, hwb = _pack(list(crops))
    engine.set_gemm_variant(1)
    r1, _ = engine.embed(pixb, offsb, hwb)
    for variant in (2, 3, 4, 5):
        engine.set_gemm_variant(variant)
        for _ in range(2):
            r2, _ = engine.embed(pixb, offsb, hwb)
            assert torch.equal(r1, r2), variant
    engine.set_gemm_variant(0)
    # the attention kernel with two LDS buffers (one head of K/V in flight) and with three (two heads) agree bit for bit
    os.environ["MME_ATTN_BUFS"] = "2"
    try:
        r3, _ = engine.embed(pixb, offsb, hwb)
    finally:
        del os.environ["MME_ATTN_BUFS"]
    assert torch.equal(r1, r3)


def test_cross_compare_api(engine):
    from multimodal_embeddings_amd.cross_compare import cross_compare

    rng = np.random.default_rng(4)
    v = rng.standard_normal((50, 768)).astype(np.float32) * 3.0
    sim = cross_compare(v.tolist(), engine=engine)
    assert isinstance(sim, np.ndarray) and sim.shape == (50, 50)
    from oracle.compare import cosine_matrix

    assert np.abs(sim - cosine_matrix(v)).max() < 1.5e-2  # bf16 rounding of the inputs
    assert np.allclose(np.diag(sim), 1.0, atol=1e-2)
    with pytest.raises(ValueError):
        cross_compare([[1.0] * 64, None], engine=engine)


def _pagesim_inputs(g, prefix):
    emb = g[f"{prefix}_emb"]
    d = emb.shape[1]
    pad = (-d) % 64
    if pad:
        emb = np.concatenate([emb, np.zeros((emb.shape[0], pad), dtype=emb.dtype)], axis=1)
    return emb, g[f"{prefix}_area_percentage"], g[f"{prefix}_page_of"]


@pytest.mark.parametrize("metric", ["cosine", "sqeuclidean"])
def test_page_similarity_vs_oracle_and_reference_golden(engine, golden_dir, metric):
    from multimodal_embeddings_amd.weighted_region_clustering import page_similarity_from_table
    from oracle import compare as ocmp

    g = np.load(os.path.join(golden_dir, "pagesim_cases.npz"))
    pages = json.load(open(os.path.join(golden_dir, "region_table.json")))
    names = [p["name"] for p in pages]
    emb, area, page_of = _pagesim_inputs(g, "real")
    assert np.all(np.diff(page_of) >= 0)
    offs = np.searchsorted(page_of, np.arange(len(names) + 1)).astype(np.int32)
    e16 = engine.normalise_rows(torch.from_numpy(emb.astype(np.float32)).cuda())
    S = page_similarity_from_table(e16, area, (area > 0).astype(np.uint8), offs, names, metric=metric, engine=engine)
    torch.cuda.synchronize()
    S = S.cpu().numpy()
    # oracle on the kernel's own cosine values: every top-k / threshold / sum decision must agree
    sims = engine.cosine(e16, e16).cpu().numpy()
    want, _ = ocmp.compute_image_similarity_matrix(None, area, page_of, names, metric=metric, sim=sims)
    assert np.abs(S - want).max() <= 1e-12
    # oracle on exact f64 dot products of the same bf16 rows: a 1e-7 difference may reorder a
    # near-tie at a top-k boundary, so only bound how many entries move
    e64 = e16.float().cpu().numpy().astype(np.float64)
    want64, _ = ocmp.compute_image_similarity_matrix(None, area, page_of, names, metric=metric, sim=e64 @ e64.T)
    assert np.mean(np.abs(S - want64) > 1e-6) <= 0.02
    assert np.array_equal(S == 0, want == 0)
    assert np.array_equal(np.diag(S), np.ones(len(names)))


def _bf16_tensor(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).cuda().view(torch.bfloat16)


@pytest.mark.parametrize("metric", ["cosine", "sqeuclidean"])
def test_page_matrix_and_labels_equal_the_reference_on_the_rows_the_device_holds(engine, golden_dir, metric):
    """GPU <-> reference DIRECTLY.  tests/golden/pagesim_bf16_cases.npz holds what the REAL
    compute_image_similarity_matrix (wrc:97-254) returned for the bf16-rounded rows themselves (inner products of the
    stored rows as the collection's cosine) and the labels of the REAL cluster_images (wrc:452-574) on that matrix.
    The device gets the same 16-bit patterns; its f32 MFMA accumulation differs from f64 by ~1e-7 per cosine, which may
    reorder a near-tie at a top-k / threshold boundary: same bound as the f64 oracle gets (<= 2 % of entries beyond
    1e-6), identical zero pattern, and the labels of K11 on the reference's matrix equal the reference's."""
    from multimodal_embeddings_amd.weighted_region_clustering import cluster_images, page_similarity_from_table

    g = np.load(os.path.join(golden_dir, "pagesim_bf16_cases.npz"))
    pages = json.load(open(os.path.join(golden_dir, "region_table.json")))
    names = [p["name"] for p in pages]
    e16 = _bf16_tensor(g["real_emb_bf16"])
    area, page_of = g["real_area_percentage"], g["real_page_of"]
    offs = np.searchsorted(page_of, np.arange(len(names) + 1)).astype(np.int32)
    S = page_similarity_from_table(e16, area, (area > 0).astype(np.uint8), offs, names, metric=metric, engine=engine).cpu().numpy()
    ref = g[f"real_S_{metric}"]
    assert np.array_equal(S == 0, ref == 0)
    assert np.mean(np.abs(S - ref) > 1e-6) <= 0.02, (np.abs(S - ref).max(), np.mean(np.abs(S - ref) > 1e-6))
    assert np.array_equal(np.diag(S), np.ones(len(names)))
    want_labels, want_k = g[f"real_labels_{metric}"].tolist(), int(g[f"real_k_{metric}"])
    res = cluster_images(ref.copy(), names, engine=engine)  # K11 on the reference's own matrix
    assert res["labels"] == want_labels and res["n_clusters"] == want_k
    res = cluster_images(S.copy(), names, engine=engine)  # and on the device's matrix: the whole chain
    assert res["labels"] == want_labels and res["n_clusters"] == want_k
    if metric == "cosine":  # the larger synthetic table (48 pages, 12..90 regions, D = 192)
        syn_names = json.load(open(os.path.join(golden_dir, "pagesim_bf16_names.json")))["names"]
        e16 = _bf16_tensor(g["syn_emb_bf16"])
        area, page_of = g["syn_area_percentage"], g["syn_page_of"]
        offs = np.searchsorted(page_of, np.arange(len(syn_names) + 1)).astype(np.int32)
        S = page_similarity_from_table(e16, area, (area > 0).astype(np.uint8), offs, syn_names, engine=engine).cpu().numpy()
        ref = g["syn_S_cosine"]
        assert np.array_equal(S == 0, ref == 0)
        assert np.mean(np.abs(S - ref) > 1e-6) <= 0.02, (np.abs(S - ref).max(), np.mean(np.abs(S - ref) > 1e-6))
        res = cluster_images(ref.copy(), syn_names, engine=engine)
        assert res["labels"] == g["syn_labels_cosine"].tolist()


def test_page_similarity_edge_cases(engine, golden_dir):
    from multimodal_embeddings_amd.weighted_region_clustering import page_similarity_from_table
    from oracle import compare as ocmp

    g = np.load(os.path.join(golden_dir, "pagesim_cases.npz"))
    names = json.load(open(os.path.join(golden_dir, "pagesim_names.json")))["names"]
    emb, area, page_of = _pagesim_inputs(g, "syn")
    types_ok = g["syn_types_ok"]
    offs = np.searchsorted(page_of, np.arange(len(names) + 1)).astype(np.int32)
    e16 = engine.normalise_rows(torch.from_numpy(emb.astype(np.float32)).cuda())
    valid = ((area > 0) & types_ok).astype(np.uint8)
    types = ["plain_text" if ok else "abandon" for ok in types_ok]
    sims = engine.cosine(e16, e16).cpu().numpy()
    for metric in ("cosine", "sqeuclidean"):
        S = page_similarity_from_table(e16, area, valid, offs, names, metric=metric, engine=engine).cpu().numpy()
        want, _ = ocmp.compute_image_similarity_matrix(None, area, page_of, names, types, metric=metric, sim=sims)
        assert np.abs(S - want).max() <= 1e-12, metric
        assert S[2, 3] == 0 and S[4].sum() == 1.0
    S = page_similarity_from_table(e16, area, valid, offs, names, skip_same_prefix=False, normalise=False, engine=engine).cpu().numpy()
    want, _ = ocmp.compute_image_similarity_matrix(None, area, page_of, names, types, skip_same_prefix=False, normalise=False, sim=sims)
    assert np.abs(S - want).max() <= 1e-15
    assert S[2, 3] > 0


def test_page_similarity_long_and_short_segments_agree_with_oracle(engine):
    """Pages of 300 regions take the general selection loop, pages of <= 256 the register-resident one; both must
    reproduce the oracle's page-pair rule on the kernel's own cosines, duplicates (ties) included."""
    from multimodal_embeddings_amd.weighted_region_clustering import page_similarity_from_table
    from oracle import compare as ocmp

    counts = [300, 7, 256, 257, 64, 1, 130, 300]
    N, d, P = sum(counts), 64, len(counts)
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(N, d, generator=g, device="cuda") + torch.randn(5, d, generator=g, device="cuda")[torch.randint(0, 5, (N,), generator=g, device="cuda")] * 1.5
    x[310:330] = x[2]  # exact duplicates inside a short page, of a row of a long one
    x[700] = x[701]
    e16 = engine.normalise_rows(x)
    rng = np.random.default_rng(4)
    area = np.exp(rng.uniform(np.log(1e-2), np.log(20.0), N))
    area[rng.random(N) < 0.03] = 0.0
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    names = [f"{p:03d} page of the segment-length test.png" for p in range(P)]
    S = page_similarity_from_table(e16, area, (area > 0).astype(np.uint8), offs, names, normalise=False, engine=engine).cpu().numpy()
    sims = engine.cosine(e16, e16).cpu().numpy()
    page_of = np.repeat(np.arange(P), counts)
    want, _ = ocmp.compute_image_similarity_matrix(None, area, page_of, names, ["plain_text"] * N, sim=sims)
    off = S[~np.eye(P, dtype=bool)]
    S = S / off.max()
    np.fill_diagonal(S, 1.0)
    assert np.abs(S - want).max() <= 1e-12
