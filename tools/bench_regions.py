#!/usr/bin/env python3
"""The reference-shaped call path at the bundled pages' geometry (PCIe inclusive, never the headline).

The 19 bundled pages (2778 x 4187 ... 7934 x 5755 pixels, 5 ... 220 embeddable boxes each, 1867 in all: tests/golden/
region_table.json; pixels seeded synthetic because the page PNGs do not ship) go from decoded host arrays to rows in the
store three ways:
  per-page   RegionProcessor.process_image_regions, one device pass per page (how an orchestrator ported 1:1 from
             complete_workflow.py:206 / region_processor.py:36-60 calls it)
  waves      RegionProcessor.process_regions: boxes of several pages per device pass, host side pipelined
  from-host  the same crops cut on the host and handed to get_image_embeddings(as_array=True) (bench.py's value_from_host shape)
each once over the 19 pages and sustained over REPS cycles of them (distinct page names, so the store grows as it would).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch

    from multimodal_embeddings_amd.embedder import RegionEmbedder
    from multimodal_embeddings_amd.region_processor import RegionProcessor, region_rows
    from multimodal_embeddings_amd.weighted_region_clustering import RegionCollection

    reps = int(os.environ.get("BENCH_REGIONS_REPS", "6"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = json.load(open(os.path.join(root, "tests", "golden", "region_table.json")))
    rng = np.random.default_rng(0)
    pages, regs, names = {}, {}, []
    t0 = time.perf_counter()
    for p in table:
        path = "/pages/" + p["name"]
        pages[path] = rng.integers(0, 256, (p["height"], p["width"], 3), dtype=np.uint8)
        r = {k: p[k] for k in ("boxes", "classes", "class_names", "scores")}
        r["image_size"] = {"width": p["width"], "height": p["height"]}
        regs[path] = r
        names.append(path)
    n_regions = sum(len(region_rows(p, regs[p])[0]) for p in names)
    page_mb = sum(a.nbytes for a in pages.values()) / 1e6
    print(f"{len(names)} pages, {n_regions} embeddable regions, {page_mb:.0f} MB of page pixels (generated in {time.perf_counter() - t0:.1f} s)", flush=True)
    emb = RegionEmbedder()
    RegionProcessor.WAVE_CROPS = int(os.environ.get("BENCH_REGIONS_WAVE", RegionProcessor.WAVE_CROPS))
    RegionProcessor.WAVE_SLOTS = int(os.environ.get("BENCH_REGIONS_SLOTS", RegionProcessor.WAVE_SLOTS))
    print(f"WAVE_CROPS {RegionProcessor.WAVE_CROPS}, WAVE_SLOTS {RegionProcessor.WAVE_SLOTS}", flush=True)
    out = {"pages": len(names), "regions": n_regions, "page_megabytes": page_mb, "reps": reps}

    def cycle(k):  # the 19 pages under new names: same pixels and boxes, new ids
        ps, rs, order = {}, {}, []
        for c in range(k):
            for p in names:
                q = p.replace("/pages/", f"/pages/cycle{c:02d} ")
                ps[q], rs[q] = pages[p], regs[p]
                order.append(q)
        return ps, rs, order

    def timed(label, fn, n):
        fn()  # warm: staging buffers, streams, allocator
        torch.cuda.synchronize()
        t = time.perf_counter()
        got = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        assert got == n, (label, got, n)
        out[label] = {"crops_per_s": n / dt, "ms": dt * 1e3, "crops": n}
        print(f"{label:34s} {n / dt:9.0f} crops/s  ({dt * 1e3:7.1f} ms for {n})", flush=True)

    rp = RegionProcessor(emb, RegionCollection())
    timed("per-page process_image_regions", lambda: sum(rp.process_image_regions(p, regs[p], page=pages[p]) for p in names), n_regions)
    for as_lists in (True, False):
        tag = "float lists" if as_lists else "ndarray rows"
        rp = RegionProcessor(emb, RegionCollection())
        timed(f"process_regions x1, {tag}", lambda: rp.process_regions(names, regions_by_path=regs, pages=pages, as_lists=as_lists), n_regions)
        ps, rs, order = cycle(reps)
        rp = RegionProcessor(emb, RegionCollection())
        timed(f"process_regions x{reps}, {tag}", lambda: rp.process_regions(order, regions_by_path=rs, pages=ps, as_lists=as_lists), n_regions * reps)
        rp.trace = {}
        rp.process_regions(order, regions_by_path=rs, pages=ps, as_lists=as_lists)
        print("   stage seconds of one more run:", {k: round(v, 3) for k, v in rp.trace.items()}, flush=True)
        rp.trace = None
    # the same crops, cut on the host, through get_image_embeddings (what bench.py reports as value_from_host)
    crops = []
    for p in names:
        _, _, boxes = region_rows(p, regs[p])
        crops += [np.ascontiguousarray(pages[p][max(y0, 0):y1, max(x0, 0):x1]) for x0, y0, x1, y1 in boxes if x1 > x0 and y1 > y0]
    timed("get_image_embeddings x1 (host crops)", lambda: int(emb.get_image_embeddings(crops, batch_size=128, as_array=True)[1].sum()), len(crops))
    many = crops * reps
    timed(f"get_image_embeddings x{reps} (host crops)", lambda: int(emb.get_image_embeddings(many, batch_size=128, as_array=True)[1].sum()), len(many))
    ref = out[f"get_image_embeddings x{reps} (host crops)"]["crops_per_s"]
    out["process_regions_frac_of_from_host"] = out[f"process_regions x{reps}, float lists"]["crops_per_s"] / ref
    out["process_regions_ndarray_frac_of_from_host"] = out[f"process_regions x{reps}, ndarray rows"]["crops_per_s"] / ref
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
