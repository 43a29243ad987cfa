"""K12 ranked neighbour lists through the C ABI (mme_neighbours) against the oracle's selection loop
(oracle/compare.py:neighbour_lists, pinned by the reference's own picks in tests/test_oracle_pins.py)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from multimodal_embeddings_amd._lib import Engine

    return Engine(0)


def _unit_bf16(engine, n, d, seed, clusters=0, dup=()):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(n, d, generator=g, device="cuda")
    if clusters:
        c = torch.randn(clusters, d, generator=g, device="cuda") * 1.5
        x = x + c[torch.randint(0, clusters, (n,), generator=g, device="cuda")]
    for a, b in dup:
        x[a] = x[b]
    return engine.normalise_rows(x)


def _check(engine, e16, group, *, fetch, top_n, rows=None, **kw):
    """kernel lists == oracle loop run on the kernel's own cosine values (identical numbers, so ties and
    window edges break the same way)."""
    from oracle import compare as oc

    n = e16.shape[0]
    row0, nrows = (0, n) if rows is None else rows
    idx, sim = engine.neighbours(e16, group, row0=row0, nrows=nrows, fetch=fetch, top_n=top_n, **kw)
    torch.cuda.synchronize()
    idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
    C = engine.cosine(e16[row0 : row0 + nrows], e16).cpu().numpy()
    want_idx, want_sim, _ = oc.neighbour_lists(None, group, top_n=top_n, fetch=fetch, sim=C, rows=range(row0, row0 + nrows), **kw)
    assert np.array_equal(idx, want_idx), np.argwhere(idx != want_idx)[:5]
    assert np.array_equal(sim, np.where(want_idx >= 0, want_sim, 0.0).astype(np.float32))
    return idx, sim


@pytest.mark.parametrize("n,d,fetch,top_n", [(1000, 768, 30, 10), (37, 64, 30, 10), (2049, 128, 100, 33), (515, 64, 128, 128), (260, 64, 1, 1)])
def test_neighbours_match_oracle_loop(engine, n, d, fetch, top_n):
    e16 = _unit_bf16(engine, n, d, seed=n, clusters=7, dup=[(3, 4), (10, 4), (n - 1, 0)])
    group = (np.arange(n) * 7 // 50).astype(np.int32)
    _check(engine, e16, group, fetch=fetch, top_n=top_n)
    _check(engine, e16, None, fetch=fetch, top_n=top_n, keep_self=True)
    _check(engine, e16, group, fetch=fetch, top_n=top_n, min_sim=0.05, max_sim=0.6)


def test_neighbours_ties_break_on_index(engine):
    """Rows that are exact copies give exactly equal cosines: order must be index ascending (stable argsort)."""
    e16 = _unit_bf16(engine, 300, 64, seed=1)
    e16[100:140] = e16[7]
    e16[200] = e16[7]
    idx, sim = _check(engine, e16, None, fetch=64, top_n=64)
    # row 7's nearest are its 41 copies, in index order, at one shared value
    assert idx[7, :41].tolist() == list(range(100, 140)) + [200] and len(set(sim[7, :41].tolist())) == 1


def test_neighbours_row_shards_equal_full(engine):
    e16 = _unit_bf16(engine, 1500, 64, seed=2, clusters=5)
    group = (np.arange(1500) // 30).astype(np.int32)
    full_idx, full_sim = engine.neighbours(e16, group, fetch=30, top_n=10)
    for row0, nrows in [(0, 700), (700, 800), (1499, 1), (5, 0)]:
        i, s = engine.neighbours(e16, group, row0=row0, nrows=nrows, fetch=30, top_n=10)
        assert torch.equal(i, full_idx[row0 : row0 + nrows]) and torch.equal(s, full_sim[row0 : row0 + nrows])


def test_neighbours_rejects_bad_arguments(engine):
    from multimodal_embeddings_amd._lib import MmeError

    e16 = _unit_bf16(engine, 64, 64, seed=3)
    for kw in [dict(fetch=0), dict(fetch=129), dict(top_n=0), dict(row0=60, nrows=10)]:
        with pytest.raises(MmeError):
            engine.neighbours(e16, None, **kw)


def test_region_report_through_reference_api(engine, golden_dir):
    """region_neighbours (mirror of region_compare.py:25) on the reference's own case: the picks the
    reference made on a brute-force f64 store, reproduced from bf16 rows on the GPU."""
    from multimodal_embeddings_amd.region_compare import region_neighbours
    from multimodal_embeddings_amd.weighted_region_clustering import RegionCollection
    from oracle import compare as oc

    g = np.load(os.path.join(golden_dir, "neighbour_cases.npz"))
    emb, page, area = g["region_emb"], g["region_page"], g["region_area"]
    n = len(emb)
    emb64 = np.concatenate([emb, np.zeros((n, 32), np.float32)], axis=1)  # D % 64 == 0, cosines unchanged
    no_box = int(g["region_no_box"][0])
    col = RegionCollection()
    ids = [f"region_{r}" for r in range(n)]
    metas = [{"parent_image": f"/data/pages/Paper {int(page[r]):02d}.png", "region_type": "plain_text", "box_str": "0,0,1,1",
              "area_percentage": float(area[r]), "is_region": True} for r in range(n)]
    del metas[no_box]["box_str"]
    col.upsert(ids=ids, embeddings=emb64.tolist(), metadatas=metas)
    rep = region_neighbours(col, score="reference_distance", engine=engine)
    assert [r["id"] for r in rep] == [i for k, i in enumerate(ids) if k != no_box]  # the box-less region is skipped as a source
    want_idx, want_d = g["region_idx"], g["region_distance"]
    same = total = 0
    for r in rep:
        k = int(r["id"].split("_")[1])
        got = [int(s["id"].split("_")[1]) for s in r["similar_regions"]]
        want = [int(c) for c in want_idx[k] if c >= 0]
        total += len(want)
        same += len(set(got) & set(want))
        for s in r["similar_regions"]:
            c = int(s["id"].split("_")[1])
            assert s["score"] >= 0.3 - 2e-2 and page[c] != page[k] and c != k
            assert s["weighted_score"] == s["score"] * (area[k] / 100) * (area[c] / 100)
            assert s["parent_image"] == f"Paper {int(page[c]):02d}.png" and s["type"] == "plain_text"
    assert same / total > 0.93, same / total  # bf16 rows move a few near-ties of the f64 ranking
    # and decision-for-decision against the oracle on the kernel's own cosines
    e16 = engine.normalise_rows(torch.from_numpy(emb64).cuda())
    _check(engine, e16, page.astype(np.int32), fetch=30, top_n=10, max_sim=float(1.0 - 0.3))
    # similarity mode: scores descend and respect the threshold
    rep2 = region_neighbours(col, score="cosine", engine=engine)
    for r in rep2:
        sc = [s["score"] for s in r["similar_regions"]]
        assert sc == sorted(sc, reverse=True) and all(x >= 0.3 for x in sc)


def test_image_report_through_reference_api(engine, golden_dir):
    from multimodal_embeddings_amd.cross_compare import image_neighbours

    g = np.load(os.path.join(golden_dir, "neighbour_cases.npz"))
    names = json.load(open(os.path.join(golden_dir, "neighbour_names.json")))["image_names"]
    rep = image_neighbours(g["image_emb"], names, engine=engine)
    want = g["image_idx"]
    same = sum(len({e["index"] for e in rep[r]} & set(want[r].tolist())) for r in range(len(names)))
    assert same / (want >= 0).sum() > 0.93
    for r, lst in enumerate(rep):
        plen = max(1, int(len(names[r]) * 0.2))
        assert len(lst) == 5 and all(e["index"] != r and e["filename"][:plen] != names[r][:plen] for e in lst)


def test_neighbours_full_size_sampled(engine):
    """C4-sized problem on one GPU: 65536 rows x 65536 candidates, multi-chunk workspace; a sample of rows
    against the oracle loop on the kernel's own cosine rows, plus structural checks on all rows."""
    from oracle import compare as oc

    n, d = 65536, 768
    e16 = _unit_bf16(engine, n, d, seed=9, clusters=64)
    group = (np.arange(n) // 128).astype(np.int32)
    gt = torch.from_numpy(group).cuda()
    idx, sim = engine.neighbours(e16, gt, fetch=30, top_n=10)
    torch.cuda.synchronize()
    assert idx.shape == (n, 10) and (idx >= 0).all()  # 30 candidates minus self and < 29 same-page rows always leave 10? checked below
    assert (sim[:, :-1] >= sim[:, 1:]).all()
    rows = torch.arange(n, device="cuda")[:, None]
    assert (idx != rows).all() and (gt[idx.long()] != gt[:, None]).all()
    sample = [0, 1, 4095, 4096, 32767, 65535]
    C = engine.cosine(e16[sample], e16).cpu().numpy()
    for k, r in enumerate(sample):
        wi, ws, _ = oc.neighbour_lists(None, group, top_n=10, fetch=30, sim=C[k : k + 1], rows=range(r, r + 1))
        assert idx[r].cpu().numpy().tolist() == wi[0].tolist() and np.array_equal(sim[r].cpu().numpy(), ws[0].astype(np.float32))


def test_neighbours_sharded_helper_single_rank_and_emulated_ranks(engine):
    """dist.neighbours_sharded: world 1 is the plain call; emulated ranks (gather=False) tile the full table."""
    from multimodal_embeddings_amd import dist as mdist

    e16 = _unit_bf16(engine, 1001, 64, seed=5, clusters=4)
    group = (np.arange(1001) // 13).astype(np.int32)
    full_idx, full_sim = engine.neighbours(e16, group, fetch=30, top_n=10)
    i1, s1 = mdist.neighbours_sharded(engine, e16, group, fetch=30, top_n=10)
    assert torch.equal(i1, full_idx) and torch.equal(s1, full_sim)
    parts = [mdist.neighbours_sharded(engine, e16, group, rank=r, world=8, gather=False, fetch=30, top_n=10) for r in range(8)]
    assert torch.equal(torch.cat([p[0] for p in parts]), full_idx) and torch.equal(torch.cat([p[1] for p in parts]), full_sim)


@pytest.mark.parametrize("kind", ["clustered", "near_duplicates", "wide_fetch"])
def test_fused_and_block_forms_are_identical(engine, kind):
    """K12 fused form (sampled threshold -> GEMM epilogue appends candidates -> selection over the lists) against
    the block form (cosine block through the workspace): same indices, same similarities, bit for bit --
    including when candidate lists overflow (near-duplicate rows) and the device re-runs the chunk unfused."""
    n, d = 20000, 128
    g = torch.Generator(device="cuda").manual_seed(77)
    if kind == "near_duplicates":
        centres = torch.randn(3, d, generator=g, device="cuda")
        x = centres[torch.randint(0, 3, (n,), generator=g, device="cuda")] + 1e-3 * torch.randn(n, d, generator=g, device="cuda")
        x[500:900] = x[17]  # exact copies as well
    else:
        centres = torch.randn(50, d, generator=g, device="cuda") * 1.5
        x = torch.randn(n, d, generator=g, device="cuda") + centres[torch.randint(0, 50, (n,), generator=g, device="cuda")]
    e16 = engine.normalise_rows(x)
    group = torch.from_numpy((np.arange(n) // 37).astype(np.int32)).cuda()
    kw = dict(fetch=128, top_n=128, keep_self=True) if kind == "wide_fetch" else dict(fetch=30, top_n=10, min_sim=0.1)
    try:
        engine.set_neighbour_mode("block")
        bi, bs = engine.neighbours(e16, None if kind == "wide_fetch" else group, **kw)
        engine.set_neighbour_mode("fused")
        fi, fs = engine.neighbours(e16, None if kind == "wide_fetch" else group, **kw)
        torch.cuda.synchronize()
        assert torch.equal(fi, bi) and torch.equal(fs, bs)
        # a shard of the rows in the fused form
        si, ss = engine.neighbours(e16, None if kind == "wide_fetch" else group, row0=4096, nrows=12000, **kw)
        assert torch.equal(si, bi[4096:16096]) and torch.equal(ss, bs[4096:16096])
    finally:
        engine.set_neighbour_mode("auto")


def test_fused_form_refuses_problems_it_cannot_take(engine):
    from multimodal_embeddings_amd._lib import MmeError

    e16 = _unit_bf16(engine, 3000, 64, seed=8)
    engine.set_neighbour_mode("fused")
    try:
        with pytest.raises(MmeError):
            engine.neighbours(e16, None)
    finally:
        engine.set_neighbour_mode("auto")
    assert engine.neighbours(e16, None)[0].shape == (3000, 10)


def test_collection_query_matches_the_reference_helper(engine, golden_dir):
    """`RegionCollection.query` through the mirror of the reference's `safe_query`: chroma's
    lists-of-lists shape, `where` filters of the three call sites, distance ascending, ties in insertion order.
    Expected: (a) what the REAL safe_query obtained from a brute-force f64 store on the same data
    (tests/golden/query_cases.json) -- same ids up to the near-ties bf16 rows move, distances to bf16 precision;
    (b) decision for decision, a f64 ranking of the SAME unit bf16 rows the kernel ranks (ids exact, distances to f32)."""
    from multimodal_embeddings_amd.weighted_region_clustering import RegionCollection, safe_query

    g = json.load(open(os.path.join(golden_dir, "query_cases.json")))
    col = RegionCollection(engine=engine)
    col.upsert(ids=g["ids"], embeddings=g["embeddings"], documents=g["documents"], metadatas=g["metadatas"])
    emb = np.asarray(g["embeddings"], dtype=np.float32)
    rows16 = engine.normalise_rows(torch.from_numpy(emb).cuda()).float().cpu().numpy().astype(np.float64)
    same = total = 0
    for case in g["cases"]:
        q = g["queries"][case["query"]]
        res = safe_query(col, q, case["n_results"], case["where"])
        assert set(res) >= {"ids", "distances", "metadatas", "documents"} and len(res["ids"]) == 1
        ids, dist = res["ids"][0], res["distances"][0]
        assert len(ids) == len(case["ids"]) == len(dist) == len(res["metadatas"][0]) == len(res["documents"][0])
        if not ids:
            continue
        assert dist == sorted(dist)
        pos = {i: k for k, i in enumerate(g["ids"])}
        assert res["metadatas"][0] == [g["metadatas"][pos[i]] for i in ids] and res["documents"][0] == [g["documents"][pos[i]] for i in ids]
        # (a) the reference helper's own result on the f64 store
        total += len(ids)
        same += len(set(ids) & set(case["ids"]))
        want_d = dict(zip(case["ids"], case["distances"]))
        assert all(abs(d - want_d[i]) <= 1e-2 for i, d in zip(ids, dist) if i in want_d)
        # (b) f64 ranking of the kernel's operands
        q16 = engine.normalise_rows(torch.tensor([q], dtype=torch.float32).cuda()).float().cpu().numpy().astype(np.float64)[0]
        if case["where"]:
            (key, cond), = case["where"].items()
            cand = [r for r, m in enumerate(g["metadatas"]) if m.get(key) == cond["$eq"]]
        else:
            cand = list(range(len(g["ids"])))
        d64 = 1.0 - rows16[cand] @ q16
        d32 = (1.0 - (rows16[cand] @ q16).astype(np.float32).astype(np.float64))
        order = np.argsort(d64, kind="stable")[: case["n_results"]]
        ref_ids = [g["ids"][cand[k]] for k in order]
        if ids != ref_ids:  # only values closer than the f32 accumulation error may swap
            for a, b in zip(ids, ref_ids):
                if a != b:
                    assert abs(d64[cand.index(pos[a])] - d64[cand.index(pos[b])]) <= 4e-6
        assert np.abs(np.array(dist) - np.array([d32[cand.index(pos[i])] for i in ids])).max() <= 4e-6
    assert same / total > 0.93, same / total
    # the stored vector used as query 3 sits at rows 11, 12 and 100: distance ~0, insertion order
    res = col.query(query_embeddings=[g["queries"][3]], n_results=3)
    assert res["ids"][0] == [g["ids"][11], g["ids"][12], g["ids"][100]] and max(res["distances"][0]) <= 1e-2
    # several queries at once, squared-L2 space (chroma's default), include subset
    col2 = RegionCollection(metric="sqeuclidean", engine=engine)
    col2.upsert(ids=g["ids"], embeddings=g["embeddings"], metadatas=g["metadatas"])
    many = col2.query(query_embeddings=g["queries"][:4], n_results=5, include=["distances"])
    one = [col.query(query_embeddings=[qv], n_results=5) for qv in g["queries"][:4]]
    assert many["ids"] == [o["ids"][0] for o in one] and many["metadatas"] is None
    assert np.allclose(many["distances"], 2.0 * np.array([o["distances"][0] for o in one]), atol=1e-6)
    # an upsert invalidates the device copy of the table
    col.upsert(ids=["new"], embeddings=[g["queries"][0]], metadatas=[{"is_region": True}])
    assert col.query(query_embeddings=[g["queries"][0]], n_results=1)["ids"] == [["new"]]


def test_create_cross_comparison_entry_point_matches_the_reference_picks(engine, golden_dir, tmp_path):
    """`create_cross_comparison(embedder, collection, image_paths, top_n)` (cross_compare.py:19): the reference's own
    entry point over a collection, against the picks the REAL function made on the same store (neighbour_cases.npz:
    parsed from the HTML pages it wrote).  The store also holds region rows (no `image_path`: the reference skips them,
    :163-166); one image is missing from the store and is re-embedded through the embedder (:94-106); an image whose
    re-embedding fails is skipped."""
    from multimodal_embeddings_amd.cross_compare import create_cross_comparison
    from multimodal_embeddings_amd.weighted_region_clustering import RegionCollection

    g = np.load(os.path.join(golden_dir, "neighbour_cases.npz"))
    names = json.load(open(os.path.join(golden_dir, "neighbour_names.json")))["image_names"]
    emb, want, want_d = g["image_emb"], g["image_idx"], g["image_distance_4dp"]
    os.makedirs(tmp_path / "imgs")
    paths = [str(tmp_path / "imgs" / nm) for nm in names]
    for p in paths:
        open(p, "w").close()  # candidates must exist on disk (:163-166)
    col = RegionCollection(engine=engine)
    missing = 11
    keep = [r for r in range(len(names)) if r != missing]
    col.add(ids=[f"image_{names[r]}" for r in keep], embeddings=[emb[r].tolist() for r in keep],
            metadatas=[{"image_name": names[r], "image_path": paths[r], "is_region": False} for r in keep], documents=[f"Image: {names[r]}" for r in keep])
    col.add(ids=["region_x_0", "region_x_1"], embeddings=[emb[0].tolist(), emb[20].tolist()],
            metadatas=[{"is_region": True, "parent_image_name": "x"}] * 2, documents=[None, None])

    class OneShotEmbedder:  # the reference's embedder contract: list in, list of float lists (or None holes) out
        calls = []

        def get_image_embeddings(self, image_paths, is_query=False, batch_size=16):
            self.calls.append(list(image_paths))
            return [emb[missing].tolist() if os.path.basename(p) == names[missing] else None for p in image_paths]

    ghost = str(tmp_path / "imgs" / "Not In The Store.png")
    out_json = tmp_path / "report" / "cross.json"
    rep = create_cross_comparison(OneShotEmbedder(), col, paths + [ghost], output_path=str(out_json))
    assert OneShotEmbedder.calls == [[paths[missing]], [ghost]]
    assert [p["image"] for p in rep] == names  # the ghost got no page; order = image_paths order
    assert col.get(ids=[f"image_{names[missing]}"])["metadatas"][0]["image_path"] == os.path.abspath(paths[missing])
    doc = json.load(open(out_json))
    assert doc["top_n"] == 5 and len(doc["images"]) == len(names)
    same = total = 0
    for r, page in enumerate(rep):
        plen = max(1, int(len(names[r]) * 0.2))
        got = [names.index(s["filename"]) for s in page["similar"]]
        assert len(got) == 5 and r not in got and all(names[c][:plen] != names[r][:plen] for c in got)
        assert all(not s["id"].startswith("region_") for s in page["similar"])
        sc = [s["score"] for s in page["similar"]]
        assert sc == sorted(sc)  # distances ascend
        for c, d in zip(got, sc):
            if c in want[r].tolist():
                same += 1
                k = want[r].tolist().index(c)
                assert abs(d - want_d[r, k]) <= 6e-3  # bf16 rows at D = 64 against the f32 store of the golden run
        total += int((want[r] >= 0).sum())
    assert same / total > 0.93  # as test_image_report_through_reference_api: near-ties reorder under bf16 rounding
