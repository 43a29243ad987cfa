#!/usr/bin/env python3
"""Headline benchmark: region-crops/sec embedded + all-pairs cosine (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N rank processes itself
(fresh children, before this process imports torch or touches a GPU) and relays rank 0's JSON
line; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
ranks come from the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).

One step = one pass of the hot path over this rank's batch of synthetic crops, all inputs
already resident in HBM: K1 crop->normalise->patchify, ViT-B/16 bf16 MFMA forward, pooling
+ L2 (K2-K8), [N>1: ONE all-gather of the bf16 embedding shards over RCCL], K9 cosine of the
local rows against the whole table.

  config c2 (default at N = 1): 4096 synthetic 224x224x3 crops, cosine [4096 x 4096]
  config c4 (default at N > 1, C4): 8192 crops per rank; the cosine block of a rank is
            [8192 x 8192*N] -- [8192 x 65536] at N = 8.  On fewer than 8 ranks `--config c4`
            pads the gathered table with seeded synthetic unit rows to 65536 (the shards the
            missing ranks would have sent), so one rank's full C4 share runs on one GPU.

  config c3 (C3): 4096 variable-size crops per GPU (the size distribution of the reference's bundled region crops,
            synthetic pixels) through the on-GPU resize / normalise / patchify + embed; `preprocess_hbm` is K1 on
            variable-size crops; a sample of rows is checked against the oracle.
  config c5 (C5): 65536 synthetic crops in all (65536 / N per rank) -> embed -> one all-gather -> cosine row block ->
            the weighted page matrix of 512 pages x 128 regions (page pairs sharded, one all-reduce) -> clustering;
            the line carries stage times and `labels_equal_oracle`.  (`--crops` shrinks it for a rehearsal.)

Weights: seeded synthetic ViT-B/16 (no checkpoint can be fetched offline).  Every line carries `table_row`, the row
of the table BASELINE.md section 4 specifies for its config and GPU count.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the bf16 MFMA GEMM,
gemm_bf16_tn_*): algorithmic GEMM FLOPs of a step / the GEMM kernels' summed duration,
measured with HIP events on the launch stream inside the timed region.  `cpu_baseline`
(N = 1 only) times the oracle's reference-shaped path on the host cores on bounded samples.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

T_IMPORT = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2_CROPS = 4096
C4_CROPS = 8192
C4_TABLE = 65536
C5_CROPS = 65536
FLOP_FORWARD_PER_CROP = 35_126_083_584  # SURVEY.md 8d (2*MAC, LN/softmax/GELU excluded)
FLOP_GEMM_PER_CROP = 231_211_008 + 12 * (697_171_968 + 232_390_656 + 2 * 929_562_624)  # K2,K4,K6,K7 launches
MFMA_BF16_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md (dense)
HBM_PEAK_GBS = 8000.0
K1_BYTES_PER_CROP = 224 * 224 * 3 + 196 * 768 * 2  # 451,584 B at 224x224
# algorithmic HBM bytes of the GEMM launches per crop: every activation operand read once, every
# output written once (bf16), residual read once; weights (85.8 M bf16 per forward pass) added per pass
BYTES_GEMM_PER_CROP = 12 * 197 * 2 * ((768 + 2304) + 3 * 768 + (768 + 3072) + (3072 + 2 * 768)) + (196 + 197) * 768 * 2
BYTES_GEMM_WEIGHTS = 2 * (768 * 768 + 12 * (2304 * 768 + 768 * 768 + 2 * 3072 * 768))
PROFILE_RECORD = os.path.join(ROOT, "profiles", "current.json")  # written by tools/profile_record.py from PMC passes of THIS command


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["auto", "c2", "c3", "c4", "c5", "tilevit"], default="auto",
                    help="auto: c2 on one GPU, c4 on several; tilevit: SURVEY 8f-2, the Mllama vision tower geometry on four-tile crops")
    ap.add_argument("--crops", type=int, default=0, help="crops per GPU (overrides the config's 4096 / 8192)")
    ap.add_argument("--table-rows", type=int, default=-1, help="rows of the table the cosine block runs against "
                    "(-1: by config; rows beyond the gathered shards are seeded synthetic unit rows)")
    ap.add_argument("--chunk", type=int, default=0, help="crops per encoder pass (0 = library default)")
    ap.add_argument("--gemm-variant", type=int, default=0, help="0 auto, 1 128x128, 2 256x256 2-slot ring, 3 256x256 3-deep activation ring")
    ap.add_argument("--no-ln-fusion", action="store_true", help="separate LayerNorm kernel instead of folding it into the GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-from-host", action="store_true", help="skip the secondary PCIe-inclusive measurement (value_from_host)")
    ap.add_argument("--no-c4-share", action="store_true", help="skip the secondary 8192-crop C4-share measurement of the N = 1 line")
    ap.add_argument("--headline-only", action="store_true", help="= --no-cpu-baseline --no-from-host --no-c4-share (profiler passes)")
    ap.add_argument("--force-dist", action="store_true", help="create the torch.distributed process group (nccl = RCCL on a GPU box) at "
                    "WORLD_SIZE = 1 too and run every collective of the N > 1 path through it: the one-GPU rehearsal of the multi-GPU code")
    ap.add_argument("--timeout", type=float, default=570.0, help="seconds the self-started multi-rank run may take before every rank is stopped (0: unbounded)")
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv, timeout_s: float = 0.0) -> int:
    """Start n fresh rank processes of this script (torchrun's environment contract) and relay them.

    Runs before torch is imported: this parent never initialises a GPU, and nothing is exec'ed from a
    process that has (the children are new interpreters).  Rank 0's stdout (the one JSON line) passes
    through -- only its JSON line: a backend that chats on stdout (gloo prints its connection banner there) must not
    break the one-line contract; the other ranks' stdout goes to stderr.  Replaces the reference's in-process device
    fan-out (deprecated_package/embedder.py:191-224) with one process per GPU.

    Every rank is polled while rank 0 is relayed: the first non-zero exit stops the others at once and becomes the return
    code (a rank that dies would otherwise leave the rest inside the collective); `timeout_s` > 0 bounds the whole run
    (return code 124)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))

    def relay():  # rank 0's stdout, on a thread of its own: the main thread must keep watching EVERY rank meanwhile
        for line in procs[0].stdout:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()

    def stop_all():
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_kill = time.monotonic() + 5.0
        for q in procs:
            try:
                q.wait(timeout=max(0.0, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()  # exactly the children started above, by handle: never by pattern

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    deadline = time.monotonic() + timeout_s if timeout_s and timeout_s > 0 else None
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:  # one rank died: the others would sit in the collective until RCCL's watchdog fires
                r, c = bad[0]
                print(f"bench.py: rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr, flush=True)
                stop_all()
                rc = c if c > 0 else 128 - c  # a signal's negative code reported the shell's way
                break
            if all(c == 0 for c in codes):
                break
            if deadline is not None and time.monotonic() > deadline:
                print(f"bench.py: --timeout {timeout_s:g} s exceeded; stopping all ranks", file=sys.stderr, flush=True)
                stop_all()
                rc = 124
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all()
        rc = 130
    reader.join(timeout=5.0)
    return rc


def profile_record():
    """Counter-derived figures of THIS command collected in separate rocprofv3 --pmc passes (profiles/current.json,
    written by tools/profile_record.py: which kernels' build, which passes, which clock).  None when absent.

    The record names the hash of the kernel sources it was collected on (`source_hash`, build.kernel_source_hash).
    When the sources of this tree hash differently the record is STALE: it is still printed (marked `stale: true`,
    with both hashes) but nothing of it feeds `roofline.traffic`."""
    try:
        with open(PROFILE_RECORD) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None
    from multimodal_embeddings_amd.build import kernel_source_hash

    here = kernel_source_hash()
    rec["stale"] = rec.get("source_hash") != here
    rec["source_hash_of_this_tree"] = here
    return rec


def baseline_metric_name():
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, KeyError, ValueError):
        return "region-crops/sec embedded + all-pairs cosine, 224×224, 1/2/4/8 MI355X"


def cpu_baseline(sample_crops, weights, budget_s: float = 14.0):
    """the reference-shaped CPU path via the oracle, on bounded samples.

    embed  : per-crop loop as embedder.py:104-137 (batch 1) on up to 256 crops, and a warmed batch-16 leg;
    compare: numpy f32 E @ E.T at 8192 x 8192 x 768;
    cluster: restated compute_image_similarity_matrix (wrc:97-254) on 32 pages x 64 regions and
             cluster_images (wrc:452-574) on a 128-page matrix."""
    import numpy as np
    import torch

    from oracle import cluster as oclu
    from oracle import compare as ocmp
    from oracle import preprocess as opre
    from oracle import vit as ovit

    # the GPU box gives one job a 16-CPU share per GPU whatever os.cpu_count() says; more
    # threads than that only oversubscribe
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    embs = []
    t0 = time.perf_counter()
    done = 0
    for crop in sample_crops[:256]:
        patches = opre.preprocess_to_patches(crop)[None]
        embs.append(ovit.vit_embed(patches, weights, batch=1)[0])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    e = np.stack(embs)
    # the same port at batch 16, warmed (a fairer best case for the CPU than the reference's one-crop loop)
    nb = min(16, len(sample_crops))
    batch = np.stack([opre.preprocess_to_patches(c) for c in sample_crops[:nb]])
    ovit.vit_embed(batch, weights, batch=nb)
    t1 = time.perf_counter()
    reps16 = 0
    while reps16 < 3 and (reps16 == 0 or time.perf_counter() - t1 < 4.0):
        ovit.vit_embed(batch, weights, batch=nb)
        reps16 += 1
    dt16 = (time.perf_counter() - t1) / reps16
    # compare: E @ E.T on unit rows
    rng = np.random.default_rng(3)
    n_cmp = 8192
    E = rng.standard_normal((n_cmp, 768), dtype=np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    t2 = time.perf_counter()
    S = E @ E.T
    dt_cmp = time.perf_counter() - t2
    del S
    # page matrix + clustering (restated reference functions)
    P, per = 32, 64
    Ep = E[: P * per].astype(np.float64)
    area = np.exp(rng.uniform(np.log(1e-2), np.log(20.0), P * per))
    pages = np.repeat(np.arange(P), per)
    names = [f"{p:04d} synthetic page of the cpu sample.png" for p in range(P)]
    t3 = time.perf_counter()
    ocmp.compute_image_similarity_matrix(Ep, area, pages, names)
    dt_page = time.perf_counter() - t3
    Pc = 128
    M = rng.uniform(0.0, 1.0, (Pc, Pc))
    M = (M + M.T) / 2
    np.fill_diagonal(M, 1.0)
    t4 = time.perf_counter()
    oclu.cluster_images(M, [f"p{i}" for i in range(Pc)])
    dt_clu = time.perf_counter() - t4
    return {
        "value": done / dt,
        "value_batch16": nb / dt16,
        "unit": "region-crops/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{done} synthetic 224x224x3 crops, per-crop fp32 torch-CPU ViT-B/16 forward (batch 1, as embedder.py:104); "
                  f"batch-16 leg warmed, {reps16} timed passes",
        "compare": {"what": f"numpy f32 E@E.T, {n_cmp} x {n_cmp} x 768", "seconds": dt_cmp, "pairs_per_s": n_cmp * n_cmp / dt_cmp},
        "page_matrix": {"what": f"oracle compute_image_similarity_matrix, {P} pages x {per} regions ({P * (P - 1) // 2} page pairs)",
                        "seconds": dt_page, "page_pairs_per_s": (P * (P - 1) // 2) / dt_page},
        "cluster": {"what": f"oracle cluster_images (average linkage + silhouette k=2..10), P = {Pc}", "seconds": dt_clu},
    }, e


def c3_inputs(n, dev):
    """Config C3: the size distribution of the reference's 1862 bundled region crops (tests/golden/
    bundled_crop_sizes_hw.npy: a data file, sorted-filename order) cycled to n crops; pixels are seeded synthetic bytes
    (the crops themselves cannot travel to the GPU box), packed like RegionEmbedder.pack (16-byte aligned)."""
    import numpy as np
    import torch

    sizes = np.load(os.path.join(ROOT, "tests", "golden", "bundled_crop_sizes_hw.npy"))
    hw = sizes[np.arange(n) % len(sizes)].astype(np.int32)
    nbytes = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
    offs = np.zeros(n, dtype=np.int64)
    offs[1:] = np.cumsum((nbytes[:-1] + 15) // 16 * 16)
    total = int(offs[-1] + nbytes[-1]) + 16
    g = torch.Generator(device=dev).manual_seed(0)
    pix = torch.randint(0, 256, (total,), dtype=torch.uint8, device=dev, generator=g)
    return pix, offs, hw, nbytes


def main():
    args = parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_from_host = args.no_c4_share = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.timeout))

    import numpy as np
    import torch
    import torch.distributed as dist

    from multimodal_embeddings_amd import dist as mdist
    from multimodal_embeddings_amd._lib import Engine
    from multimodal_embeddings_amd.weights import make_vit_weights, synthetic_crops, synthetic_page_structure

    rank, world, local = mdist.init_from_env(force=args.force_dist)
    use_dist = mdist.collectives_active()  # world > 1, or the forced process group of one rank
    if args.config == "tilevit":
        return main_tilevit(args, rank, world, local, use_dist)
    if world != args.gpus:
        if rank == 0:
            print(f"error: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or run "
                  f"`python bench.py --gpus {args.gpus}` without a launcher (it starts its own ranks)", file=sys.stderr)
        sys.exit(2)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    config = args.config if args.config != "auto" else ("c2" if world == 1 else "c4")
    if config == "c5":
        n = args.crops or C5_CROPS // world
        table_rows = n * world
    elif config == "c3":
        n = args.crops or C2_CROPS
        table_rows = n * world
    else:
        n = args.crops or (C2_CROPS if config == "c2" else C4_CROPS)
        table_rows = args.table_rows if args.table_rows >= 0 else (max(C4_TABLE, n * world) if config == "c4" and not args.crops else n * world)
        table_rows = max(table_rows, n * world)
    n_synth = table_rows - n * world

    weights = make_vit_weights(seed=1)
    eng = Engine(local)
    eng.load_vit(weights)
    if args.chunk:
        eng.set_chunk(args.chunk)
    if args.gemm_variant:
        eng.set_gemm_variant(args.gemm_variant)
    if args.no_ln_fusion:
        eng.set_ln_fusion(False)

    start = rank * n
    bytes_in = None
    if config == "c3":
        pix, offs, hw, nbytes = c3_inputs(n, dev)
        bytes_in = float(nbytes.sum())
        crops_host = None
    else:
        # synthetic 224 x 224 x 3 crops, generated in blocks so that 65536 of them never sit in host memory twice
        pix = torch.empty(n * 224 * 224 * 3 + 16, dtype=torch.uint8, device=dev)
        crops_host = None
        for b0 in range(0, n, 4096):
            blk = synthetic_crops(min(4096, n - b0), seed=0, start=start + b0)
            if b0 == 0:
                crops_host = blk
            pix[b0 * 150528: (b0 + len(blk)) * 150528] = torch.from_numpy(blk.reshape(-1)).to(dev)
        offs = np.arange(n, dtype=np.int64) * (224 * 224 * 3)
        hw = np.tile(np.array([[224, 224]], dtype=np.int32), (n, 1))
    e32 = torch.empty((n, 768), dtype=torch.float32, device=dev)
    e16 = torch.empty((n, 768), dtype=torch.bfloat16, device=dev)
    table = torch.empty((table_rows, 768), dtype=torch.bfloat16, device=dev)
    if n_synth:  # the shards of the ranks that are not there: seeded unit rows, resident before the timed region
        g = torch.Generator(device=dev).manual_seed(17)
        table[n * world:] = eng.normalise_rows(torch.randn(n_synth, 768, generator=g, device=dev))
    sim = torch.empty((n, table_rows), dtype=torch.float32, device=dev)
    gather_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 1))]
    gather_host_s = [0.0]
    c5 = None
    if config == "c5":  # page structure of SURVEY.md 8d: P pages x 128 regions over ALL ranks' crops
        per_page = 128
        if (n * world) % per_page:
            raise SystemExit(f"--config c5 needs a multiple of {per_page} crops in total, got {n * world}")
        area_pct, page_offs, page_names = synthetic_page_structure((n * world) // per_page, per_page, seed=2, duplicated_prefixes=16)
        c5 = {"S": None, "labels": None, "k": None, "stage_s": {"embed": 0.0, "gather+cosine": 0.0, "page_matrix": 0.0, "cluster": 0.0}}
        valid = np.ones(n * world, dtype=np.uint8)

    def step(i=-1):
        t_a = time.perf_counter()
        eng.embed(pix, offs, hw, 0, out_f32=e32, out_bf16=e16)
        if c5 is not None and i >= 0:
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            c5["stage_s"]["embed"] += t_b - t_a
        if use_dist:
            if i >= 0:
                gather_ev[i][0].record()
                th = time.perf_counter()
            mdist.all_gather_rows(e16, out=table[: n * world])
            if i >= 0:
                gather_ev[i][1].record()
                gather_host_s[0] += time.perf_counter() - th
        else:
            table[:n].copy_(e16)
        eng.cosine(e16, table, out=sim)
        if c5 is None:
            return
        # the compare -> cluster half of the chain (wrc:857-892): page pairs sharded over the ranks, one all-reduce of the
        # P x P f64 partials, clustering on every rank (one workgroup, milliseconds)
        if i >= 0:
            torch.cuda.synchronize()
            t_c = time.perf_counter()
            c5["stage_s"]["gather+cosine"] += t_c - t_b
        S = mdist.page_similarity_sharded(table, area_pct, valid, page_offs, page_names, rank=rank, world=world, engine=eng)
        if i >= 0:
            torch.cuda.synchronize()
            t_d = time.perf_counter()
            c5["stage_s"]["page_matrix"] += t_d - t_c
        labels, k, _ = eng.cluster_pages(S)
        if i >= 0:
            c5["stage_s"]["cluster"] += time.perf_counter() - t_d
        c5["S"], c5["labels"], c5["k"] = S, labels, k

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # seconds from the creation of this rank's process to the first step: interpreter + torch import, process group,
    # synthetic weights (a numpy hash generator, single-threaded) and crops, uploads -- N of these run side by side at N ranks
    try:
        import psutil

        startup_s = time.time() - psutil.Process().create_time()
    except Exception:  # noqa: BLE001
        startup_s = time.time() - T_IMPORT
    for _ in range(args.warmup):
        step()
    fence()
    eng.profile(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed_local = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    per_rank = mdist.all_gather_floats(elapsed_local, dev)
    startup_by_rank = mdist.all_gather_floats(startup_s, dev)
    # which class of box every rank ran on: outside the timed region.  --headline-only (the profiler passes) skips
    # the clock probe: its 0.5 s of GEMM launches would be folded into the kernel statistics of the step
    box = box_probe(eng, dev, clock=not args.headline_only)
    box_by_rank = {k: mdist.all_gather_floats(v if v is not None else float("nan"), dev) for k, v in box.items()}
    elapsed = max(per_rank)
    steps = max(args.steps, 1)
    gather_ms = None
    if use_dist and args.steps > 0:
        # device-side span of the collective on the launch stream (it includes waiting for the slowest rank's shard)
        gather_ms = max(mdist.all_gather_floats(sum(a.elapsed_time(b) for a, b in gather_ev[: args.steps]) / steps, dev))

    if rank == 0:
        ms_step = elapsed * 1e3 / steps
        total_crops = n * world * steps
        value = total_crops / elapsed
        gemm_ms, gemm_launches = prof["gemm"]
        gemm_ms_step = gemm_ms / steps
        ach = (FLOP_GEMM_PER_CROP * n) / (gemm_ms_step * 1e-3) / 1e12 if gemm_ms > 0 and c5 is None else None
        cos_ms = prof["cosine"][0] / steps if prof["cosine"][1] else None
        cos_bytes = float(n) * table_rows * 4 + float(table_rows) * 768 * 2 + float(n) * 768 * 2  # f32 block written + bf16 rows read
        cosine_hbm = {"achieved": cos_bytes / (cos_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cos_bytes / (cos_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                      "bytes": cos_bytes, "ms": cos_ms, "block": [n, table_rows]} if cos_ms else None
        pre_ms = prof["preprocess"][0] / steps if prof["preprocess"][1] else None
        k1_bytes = (bytes_in + 301056.0 * n) if bytes_in is not None else float(K1_BYTES_PER_CROP) * n
        preprocess_hbm = {"achieved": k1_bytes / (pre_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": k1_bytes / (pre_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), "bytes": k1_bytes, "ms": pre_ms} if pre_ms else None
        rec = profile_record()
        traffic = rec.get("gemm_traffic_per_launch") if rec and rec.get("crops_per_gpu") == n and not rec["stale"] and config == "c2" else None
        roofline = {
            "kernel": "gemm_bf16_tn (K2/K4/K6/K7 launches of the ViT forward)",
            "bound": "mfma",
            "achieved": ach,
            "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": (ach / MFMA_BF16_PEAK_TFLOPS) if ach else None,
            "traffic": traffic,
            "traffic_unit": "bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes of this command; see from_profile; "
                            "null when the record is stale or was collected on another workload)",
            "algorithmic_bytes_per_launch": (BYTES_GEMM_PER_CROP * n + BYTES_GEMM_WEIGHTS * (gemm_launches / steps / 49.0)) / (gemm_launches / steps) if gemm_launches else None,
            "launches_per_step": gemm_launches / steps,
            "avg_launch_ms": gemm_ms / gemm_launches if gemm_launches else None,
            "flop_per_launch_avg": FLOP_GEMM_PER_CROP * n / (gemm_launches / steps) if gemm_launches else None,
        }
        if c5 is not None:
            # C5's GEMM class also holds K9 and K10's query GEMM: price the forward from its own launches only
            roofline.update(achieved=None, frac=None, note="C5 mixes K9 / K10 launches into the GEMM class; see the C2 line for the forward's roofline")
        what = {"c2": "synthetic 224x224x3 crops", "c4": "synthetic 224x224x3 crops", "c5": "synthetic 224x224x3 crops",
                "c3": "variable-size crops (bundled size distribution, synthetic pixels)"}[config]
        workload = (f"{config.upper()}: {n} {what} per GPU -> K1 patchify + ViT-B/16 bf16 forward + pool/L2 + "
                    f"[{n} x {table_rows}] cosine; seeded synthetic weights")
        if config == "c5":
            workload += f" -> page matrix of {len(page_names)} pages x 128 regions (K10, pair shards + one all-reduce) -> clustering (K11)"
        if n_synth:
            workload += f"; {n_synth} table rows are seeded synthetic unit rows standing in for the shards of absent ranks"
        forward_frac = FLOP_FORWARD_PER_CROP * (n * steps / elapsed_local) / (MFMA_BF16_PEAK_TFLOPS * 1e12)
        out = {
            "metric": baseline_metric_name(),
            "value": value,
            "unit": "region-crops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak" if config != "c5" else "strong",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "name": config,
                "crops_per_gpu": n,
                "table_rows": table_rows,
                "parallelism": f"dp{world} (crop shards, one all-gather of bf16 embeddings, backend {dist.get_backend()}"
                               f"{'; process group forced at world size 1' if world == 1 else ''})" if use_dist else "single GPU",
                "note": "the default line is C2 (4096 crops per GPU) at N = 1 and C4's per-rank share (8192 crops per GPU, weak scaling) at N > 1; "
                        "`c4_share_at_this_n` in the N = 1 line is the same 8192-crop share measured in this run, the figure a 1 -> N curve "
                        "should be read against",
            },
            "ms_per_step_by_rank": [t * 1e3 / steps for t in per_rank],
            "startup_s_by_rank": startup_by_rank,
            "box": {**{k: v[0] for k, v in box_by_rank.items()}, "by_rank": box_by_rank,
                    "what": "hbm_copy_gbs: device-to-device copy of 1 GiB (read + written bytes / HIP-event time, ~50 ms of copies): boxes of this "
                            "pool stream at ~3.9 or ~5.5 TB/s in K1 and differ up to 30 % on the HBM-bound kernels; gemm_clock_ghz: s_memtime / "
                            "s_memrealtime of one stamped launch of the QKV-shaped GEMM after 0.5 s of the product kernel (mme_gemm_stamps)"},
            "result_digest": result_digest(e32, table[: n * world], c5),
            "allgather_ms": gather_ms,
            "allgather_host_ms": gather_host_s[0] * 1e3 / steps if use_dist else None,
            "allgather_bytes": float(n) * world * 768 * 2 if use_dist else None,
            "attention_layers_redone_last_pass": int(sum(eng.attention_redone())),  # fast softmax form's guard (mme.h): 0 = no exact re-run
            "forward_mfma_frac": forward_frac if c5 is None else None,
            "forward_mfma_frac_note": "forward FLOP x this rank's crops/s / 2.5 PFLOP/s nominal dense peak (no clock adjustment), measured in this run",
            "kernel_ms_per_step": {k: v[0] / steps for k, v in prof.items() if v[1]},
            "cosine_hbm": cosine_hbm,
            "preprocess_hbm": preprocess_hbm,
            "roofline": roofline,
            "from_profile": rec,
        }
        parity = None
        labels_equal = None
        if world == 1 and config == "c2" and not args.no_cpu_baseline:
            emb_head = e32[:256].cpu().numpy()  # of the timed steps (the secondary legs below reuse the engine)
            cb, ecpu = cpu_baseline(crops_host, weights)
            out["cpu_baseline"] = cb
            parity = float(np.max(1.0 - np.sum(emb_head[: len(ecpu)] * ecpu, axis=1)))
        elif not args.no_cpu_baseline and config in ("c3", "c5"):
            # bounded parity sample against the oracle (test infrastructure, used here as the checker only)
            from oracle import preprocess as opre
            from oracle import vit as ovit

            idx = [0, 1, 17, n // 2, n - 1]
            host = pix.cpu().numpy() if config == "c3" else None
            crops = [host[offs[i]: offs[i] + int(hw[i, 0]) * int(hw[i, 1]) * 3].reshape(int(hw[i, 0]), int(hw[i, 1]), 3) for i in idx] if config == "c3" \
                else [pix[i * 150528: (i + 1) * 150528].cpu().numpy().reshape(224, 224, 3) for i in idx]
            want = ovit.vit_embed(np.stack([opre.preprocess_to_patches(c) for c in crops]), weights)
            parity = float(np.max(1.0 - np.sum(e32[idx].cpu().numpy() * want, axis=1)))
        if c5 is not None:
            out["c5"] = c5_checks(c5, eng, table, area_pct, page_offs, page_names, steps, check=not args.no_cpu_baseline)
            labels_equal = out["c5"].get("labels_equal_oracle")
        if parity is not None:
            out["parity_max_1_minus_cos_vs_oracle"] = parity
        # secondary legs: a failure there (say, pinned host memory refused) must not cost the headline line
        if world == 1 and config == "c2" and not args.crops and not args.no_c4_share:
            try:
                out["c4_share_at_this_n"] = c4_share_line(eng, weights, dev, args)
            except Exception as e:  # noqa: BLE001
                out["c4_share_at_this_n"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and config == "c2" and not args.no_c4_share:
            try:  # the same step with the rows nothing reads left out of the LAST layer: a secondary figure, never `value`
                eng.set_forward_pruning(True)
                p32 = torch.empty_like(e32)
                eng.embed(pix, offs, hw, 0, out_f32=p32, out_bf16=e16)
                torch.cuda.synchronize()
                k = max(2, min(args.steps, 5))
                tp = time.perf_counter()
                for _ in range(k):
                    eng.embed(pix, offs, hw, 0, out_f32=p32, out_bf16=e16)
                    table[:n].copy_(e16)
                    eng.cosine(e16, table, out=sim)
                torch.cuda.synchronize()
                dtp = (time.perf_counter() - tp) / k
                out["value_last_layer_pruned"] = {
                    "value": n / dtp, "unit": "region-crops/s", "ms_per_step": dtp * 1e3, "steps": k,
                    "embeddings_bit_identical_to_the_full_pass": bool(torch.equal(p32, e32)),
                    "what": "mme_set_forward_pruning(1): in the last layer only the query block that holds the pooled token is attended and its "
                            "o_proj / LayerNorm / MLP run on the n gathered rows (6.2 % of the forward's FLOP are rows nothing reads); OFF in "
                            "the headline value, which times the whole forward"}
            except Exception as e:  # noqa: BLE001
                out["value_last_layer_pruned"] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                eng.set_forward_pruning(False)
        if world == 1 and config in ("c2", "c3") and not args.no_from_host:
            try:
                out["value_from_host"] = from_host_line(eng, config, crops_host, pix if config == "c3" else None,
                                                        offs if config == "c3" else None, hw if config == "c3" else None, value)
            except Exception as e:  # noqa: BLE001
                out["value_from_host"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and config == "c3" and not args.no_from_host:
            try:  # the reference-shaped path from decoded PAGES (region_processor.py:36-60), PCIe inclusive: never `value`
                out["value_through_process_regions"] = process_regions_line(eng, out.get("value_from_host", {}).get("value"))
            except Exception as e:  # noqa: BLE001
                out["value_through_process_regions"] = {"error": f"{type(e).__name__}: {e}"}
        # one row per config x GPU count
        out["table_row"] = {
            "config": config.upper(), "gpus": world, "crops_per_s": value,
            "forward_tflops": FLOP_FORWARD_PER_CROP * value / world / 1e12 if c5 is None else None,
            "forward_pct_of_2p5pf": 100.0 * forward_frac if c5 is None else None,
            "rocprof_mfma_util": rec.get("forward_mfma_util") if rec and not rec["stale"] else None,
            "preprocess_gbs": preprocess_hbm["achieved"] if preprocess_hbm else None, "preprocess_pct_hbm": 100.0 * preprocess_hbm["frac"] if preprocess_hbm else None,
            "cosine_gbs": cosine_hbm["achieved"] if cosine_hbm else None, "cosine_pct_hbm": 100.0 * cosine_hbm["frac"] if cosine_hbm else None,
            "allgather_ms": gather_ms,
            "cpu_crops_per_s": out.get("cpu_baseline", {}).get("value"), "cpu_cores": out.get("cpu_baseline", {}).get("cores"),
            "max_cosine_error_vs_cpu_ref": parity, "labels_equal_oracle": labels_equal,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


TILE_TOKENS = 4 * 1608  # one crop = ONE sequence of 4 tiles x 1608 tokens (padding tiles and tokens included: the model attends to them)
FLOP_TILE_ATTN_PER_LAUNCH_PER_CROP = 4 * TILE_TOKENS * TILE_TOKENS * 1280  # QK^T + PV of the 16 heads of one layer
FLOP_TILE_GEMM_PER_CROP = 40 * 2 * TILE_TOKENS * (1280 * 3840 + 1280 * 1280 + 2 * 1280 * 5120) + 2 * 6400 * 588 * 1280
FLOP_TILE_PER_CROP = 40 * FLOP_TILE_ATTN_PER_LAUNCH_PER_CROP + FLOP_TILE_GEMM_PER_CROP


def main_tilevit(args, rank, world, local, use_dist):
    """`--config tilevit`: four-tile crops through `mme_tile_vit_forward` -- the reference checkpoint's own
    vision-tower geometry (transformers `MllamaVisionModel` at image 560: 32 local + 8 gated global layers, 1280-d, 16
    heads of 80, one sequence of 6432 tokens per crop) on seeded synthetic weights.  A step = one pass over this rank's
    crops, pixel values resident; ranks are replicas (crops are independent, no collective).  `roofline` is the attention
    kernel `attn_fwd_tiles` (the largest class of the pass); `roofline_gemm` the GEMM launches; `cpu_baseline` the oracle
    (oracle/mllama_vision.py, torch-CPU f32) on ONE crop through 4 local + 1 global layers, scaled to 32 + 8."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from multimodal_embeddings_amd import dist as mdist
    from multimodal_embeddings_amd._lib import Engine
    from multimodal_embeddings_amd.embedder import RegionEmbedder
    from multimodal_embeddings_amd.weights import TileViTGeometry, make_tile_vit_weights

    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    n = args.crops or 8
    w = make_tile_vit_weights(2)
    eng = Engine(local)
    eng.load_tile_vit(w)
    rng = np.random.default_rng(rank)
    arrays = [rng.integers(0, 256, (1000 + 7 * k, 1100, 3), dtype=np.uint8) for k in range(n)]  # every crop fits the 2 x 2 tile grid
    emb = RegionEmbedder(engine=eng)
    pix, offs, hw = emb.pack(arrays, dev)
    pv, ids, _, nt = eng.preprocess_tiles(pix, offs, hw, 560, 4)
    assert all(int(v) == 4 for v in nt)
    eng.set_chunk(n)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        eng.tile_vit_forward(pv, ids, nt)
    fence()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, e32, _ = eng.tile_vit_forward(pv, ids, nt)
    fence()
    elapsed_local = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    per_rank = mdist.all_gather_floats(elapsed_local, dev)
    elapsed = max(per_rank)
    steps = max(args.steps, 1)
    if rank == 0:
        att_ms, att_l = prof["attention"]
        gemm_ms, gemm_l = prof["gemm"]
        ach_att = FLOP_TILE_ATTN_PER_LAUNCH_PER_CROP * n * att_l / (att_ms * 1e-3) / 1e12 if att_ms > 0 else None
        ach_gemm = FLOP_TILE_GEMM_PER_CROP * n * steps / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None
        out = {
            "metric": baseline_metric_name(), "value": n * world * steps / elapsed, "unit": "region-crops/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"TILEVIT: {n} four-tile crops (1000..1049 x 1100 px -> 2 x 2 tiles of 560 x 560) per GPU through the Mllama vision "
                                   "tower geometry (32 local + 8 gated global layers, 1280-d, 16 heads of 80, 6432 tokens per crop, 7680-d output); "
                                   "pixel values resident; seeded synthetic weights",
                       "name": "tilevit", "crops_per_gpu": n, "parallelism": f"replicas x{world} (no collective)" if world > 1 else "single GPU",
                       "note": "SURVEY.md 8f-2 (the reference checkpoint's own encoder geometry); NOT the configuration BASELINE.json's metric is "
                               "quoted on (that is ViT-B/16 at 224 x 224, the default line)"},
            "ms_per_step_by_rank": [t * 1e3 / steps for t in per_rank],
            "forward_tflops": FLOP_TILE_PER_CROP * n * steps / elapsed_local / 1e12,
            "forward_mfma_frac": FLOP_TILE_PER_CROP * n * steps / elapsed_local / (MFMA_BF16_PEAK_TFLOPS * 1e12),
            "flop_per_crop": FLOP_TILE_PER_CROP,
            "kernel_ms_per_step": {k: v[0] / steps for k, v in prof.items() if v[1]},
            "roofline": {"kernel": "attn_fwd_tiles (self-attention of one layer over 6432-token sequences, 16 heads of 80)", "bound": "mfma",
                         "achieved": ach_att, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_att / MFMA_BF16_PEAK_TFLOPS if ach_att else None,
                         "traffic": None, "launches_per_step": att_l / steps, "avg_launch_ms": att_ms / att_l if att_l else None,
                         "flop_per_launch": FLOP_TILE_ATTN_PER_LAUNCH_PER_CROP * n,
                         "note": "algorithmic FLOP = 4 T^2 d per layer and crop (QK^T and PV at the real head dim 80; the zero-padded third value block "
                                 "the kernel multiplies is not counted); duration = HIP events around every launch on the launch stream"},
            "roofline_gemm": {"kernel": "gemm_bf16_tn_256r (patch, QKV, o_proj, fc1, fc2 launches of the tower)", "bound": "mfma", "achieved": ach_gemm,
                              "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_gemm / MFMA_BF16_PEAK_TFLOPS if ach_gemm else None,
                              "launches_per_step": gemm_l / steps, "avg_launch_ms": gemm_ms / gemm_l if gemm_l else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import mllama_vision as omv

            torch.set_num_threads(min(os.cpu_count() or 1, 16))
            shallow = TileViTGeometry(num_layers=4, num_global_layers=1, intermediate_layers=(1, 3))
            t1 = time.perf_counter()
            want = omv.vision_forward(pv[0].cpu().numpy(), int(ids[0]), int(nt[0]), w, shallow)
            dt = time.perf_counter() - t1
            scale = 40.0 / 5.0
            out["cpu_baseline"] = {"value": 1.0 / (dt * scale), "unit": "region-crops/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"1 four-tile crop through 4 local + 1 global layers of the oracle (torch-CPU f32), {dt:.1f} s, scaled x{scale:g} "
                                             "to the 32 + 8 layers of the tower (layers cost the same)", "seconds_sample": dt}
            # the same shallow stack on the GPU against that oracle run (the checker, on the sample it just produced)
            eng2 = Engine(local)
            eng2.load_tile_vit(w, shallow)
            hid, _, _ = eng2.tile_vit_forward(pv[:1], ids[:1], nt[:1], want_hidden=True)
            got = hid[0].cpu().numpy().reshape(-1, want.shape[-1]).astype(np.float64)
            ref = want.reshape(-1, want.shape[-1]).astype(np.float64)
            cos = np.sum(got * ref, axis=1) / np.maximum(np.linalg.norm(got, axis=1) * np.linalg.norm(ref, axis=1), 1e-30)
            out["parity_min_token_cosine_vs_oracle"] = float(cos.min())
            eng2.close()
        out["table_row"] = {"config": "TILEVIT", "gpus": world, "crops_per_s": out["value"], "forward_tflops": out["forward_tflops"],
                            "forward_pct_of_2p5pf": 100.0 * out["forward_mfma_frac"], "cpu_crops_per_s": out.get("cpu_baseline", {}).get("value"),
                            "cpu_cores": out.get("cpu_baseline", {}).get("cores")}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def box_probe(eng, dev, clock=True):
    """Two numbers that tell boxes apart: the rate of a plain device copy (HBM class) and the clock the chip holds
    under the dominant GEMM.  ~0.6 s in all, after the timed region."""
    import numpy as np
    import torch

    out = {"hbm_copy_gbs": None, "gemm_clock_ghz": None}
    try:
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        src.random_(0, 256)
        for _ in range(3):
            dst.copy_(src)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 100
        a.record()
        for _ in range(reps):
            dst.copy_(src)
        b.record()
        b.synchronize()
        out["hbm_copy_gbs"] = 2.0 * nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9
        del src, dst
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: copy probe failed: {e}", file=sys.stderr)
    if not clock:
        return out
    try:
        st = eng.gemm_stamps(806912, 2304, 768).astype(np.float64)
        ok = st[:, 0, 10] > 0
        out["gemm_clock_ghz"] = float(np.median(st[ok, 0, 13] / np.maximum(st[ok, 0, 14], 1.0) * 0.1))
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: clock probe failed: {e}", file=sys.stderr)
    return out


def result_digest(e32, table_head, c5):
    """sha256 of what the step produced (rank 0's f32 embeddings, the gathered bf16 table head, and for C5 the page
    matrix and labels): two runs of the same configuration -- with and without the process group, say -- are
    bit-identical exactly when these agree."""
    import hashlib

    import torch

    def sha(t):
        return hashlib.sha256(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:32]

    out = {"embeddings_f32": sha(e32), "table_bf16": sha(table_head)}
    if c5 is not None and c5["S"] is not None:
        out["page_matrix_f64"] = sha(c5["S"])
        out["labels"] = [int(v) for v in c5["labels"]]
    return out


def c4_share_line(eng, weights, dev, args):
    """One rank's share of C4 on this GPU (8192 crops + the [8192 x 65536] block against a table whose other 57344 rows
    are seeded synthetic unit rows): what the N > 1 lines run per GPU, measured beside the C2 line so that the driver's
    1 -> N curve has an N = 1 point of the SAME per-GPU workload."""
    import numpy as np
    import torch

    from multimodal_embeddings_amd.weights import synthetic_crops

    n, rows = C4_CROPS, C4_TABLE
    pix = torch.empty(n * 150528 + 16, dtype=torch.uint8, device=dev)
    for b0 in range(0, n, 4096):
        pix[b0 * 150528: (b0 + 4096) * 150528] = torch.from_numpy(synthetic_crops(4096, seed=0, start=b0).reshape(-1)).to(dev)
    offs = np.arange(n, dtype=np.int64) * 150528
    hw = np.tile(np.array([[224, 224]], dtype=np.int32), (n, 1))
    e32 = torch.empty((n, 768), dtype=torch.float32, device=dev)
    e16 = torch.empty((n, 768), dtype=torch.bfloat16, device=dev)
    table = torch.empty((rows, 768), dtype=torch.bfloat16, device=dev)
    g = torch.Generator(device=dev).manual_seed(17)
    table[n:] = eng.normalise_rows(torch.randn(rows - n, 768, generator=g, device=dev))
    sim = torch.empty((n, rows), dtype=torch.float32, device=dev)

    def step():
        eng.embed(pix, offs, hw, 0, out_f32=e32, out_bf16=e16)
        table[:n].copy_(e16)
        eng.cosine(e16, table, out=sim)

    step()
    torch.cuda.synchronize()
    k = max(2, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    return {"value": n / dt, "unit": "region-crops/s", "crops_per_gpu": n, "table_rows": rows, "ms_per_step": dt * 1e3, "steps": k}


def from_host_line(eng, config, crops_host, pix, offs, hw, resident_value):
    """Secondary, PCIe-INCLUSIVE rate (never `value`): the same crops handed over as host numpy arrays through the
    reference-shaped entry, RegionEmbedder.get_image_embeddings (embedder.py:141-226) -- packing into pinned memory, H2D,
    device pass, D2H and the result conversion, pipelined over groups of 2048 crops.  16384 crops per timed call (the
    4096-crop set four times over), once returning an ndarray (`as_array=True`) and once the reference's list of float
    lists."""
    import numpy as np

    from multimodal_embeddings_amd.embedder import RegionEmbedder

    if config == "c3":
        host = pix.cpu().numpy()
        arrays = [host[offs[i]: offs[i] + int(hw[i, 0]) * int(hw[i, 1]) * 3].reshape(int(hw[i, 0]), int(hw[i, 1]), 3) for i in range(len(offs))]
    else:
        arrays = [crops_host[i] for i in range(len(crops_host))]
    arrays = arrays * 4
    emb = RegionEmbedder(engine=eng)
    emb.get_image_embeddings(arrays[: len(arrays) // 2], batch_size=128, as_array=True)  # warm: staging buffers, streams
    t0 = time.perf_counter()
    arr, ok = emb.get_image_embeddings(arrays, batch_size=128, as_array=True)
    dt_arr = time.perf_counter() - t0
    t0 = time.perf_counter()
    lists = emb.get_image_embeddings(arrays, batch_size=128)
    dt_list = time.perf_counter() - t0
    assert ok.all() and all(v is not None for v in lists)
    same = bool(np.array_equal(arr[: len(arrays) // 4], arr[len(arrays) // 4: len(arrays) // 2]) and np.array_equal(arr[0], np.asarray(lists[0], dtype=np.float32)))
    mb = sum(a.nbytes for a in arrays) / 1e6
    return {"value": len(arrays) / dt_arr, "unit": "region-crops/s", "crops": len(arrays), "host_megabytes": mb,
            "frac_of_resident": len(arrays) / dt_arr / resident_value,
            "value_float_lists": len(arrays) / dt_list, "frac_of_resident_float_lists": len(arrays) / dt_list / resident_value,
            "repeats_agree": same,
            "what": "host uint8 arrays -> get_image_embeddings(batch_size=128): pinned packing + H2D of group g+1 and D2H / conversion of group g-1 "
                    "under the device pass of group g; PCIe inclusive, never the headline value"}


def process_regions_line(eng, from_host_value, cycles=4):
    """Secondary figure of the C3 line: the 19 bundled pages' geometry (tests/golden/region_table.json: 1867
    embeddable boxes, 1.2 GB of page pixels; pixels seeded synthetic) from decoded host pages to rows in the store through
    `RegionProcessor.process_regions` -- page uploads, boxes cut on the device, >= 1024 crops per device pass, rows handed to the
    store per page -- over `cycles` rounds of the pages (distinct names), against the same call page by page."""
    import gc

    import numpy as np

    from multimodal_embeddings_amd.embedder import RegionEmbedder
    from multimodal_embeddings_amd.region_processor import RegionProcessor, region_rows
    from multimodal_embeddings_amd.weighted_region_clustering import RegionCollection

    table = json.load(open(os.path.join(ROOT, "tests", "golden", "region_table.json")))
    rng = np.random.default_rng(0)
    pages, regs, order = {}, {}, []
    for c in range(cycles):
        for p in table:
            path = f"/pages/cycle{c:02d} " + p["name"]
            if c == 0:
                pix = rng.integers(0, 256, (p["height"], p["width"], 3), dtype=np.uint8)
                reg = {k: p[k] for k in ("boxes", "classes", "class_names", "scores")}
                reg["image_size"] = {"width": p["width"], "height": p["height"]}
            else:
                pix, reg = pages["/pages/cycle00 " + p["name"]], regs["/pages/cycle00 " + p["name"]]
            pages[path], regs[path] = pix, reg
            order.append(path)
    n = sum(len(region_rows(p, regs[p])[0]) for p in order)
    emb = RegionEmbedder(engine=eng)
    res = {}
    for label, run in (("waves", lambda rp: rp.process_regions(order, regions_by_path=regs, pages=pages)),
                       ("page_by_page", lambda rp: sum(rp.process_image_regions(p, regs[p], page=pages[p]) for p in order[: len(table)]))):
        rp = RegionProcessor(emb, RegionCollection())
        run(rp)  # warm: staging buffers, streams
        best = 0.0
        for _ in range(2):  # best of two: the host side of this path (page uploads from pageable memory, two threads) varies run to run
            rp = RegionProcessor(emb, RegionCollection()) if label == "page_by_page" else rp
            gc.collect()
            t0 = time.perf_counter()
            got = run(rp)
            best = max(best, got / (time.perf_counter() - t0))
        res[label] = best
    return {"value": res["waves"], "unit": "region-crops/s", "crops": n, "pages": len(order), "value_page_by_page": res["page_by_page"],
            "frac_of_value_from_host": res["waves"] / from_host_value if from_host_value else None,
            "what": "decoded host pages -> RegionProcessor.process_regions (boxes of several pages per device pass) -> float-list rows in a "
                    "RegionCollection, PCIe inclusive, never the headline value; value_page_by_page = process_image_regions per page"}


def c5_checks(c5, eng, table, area_pct, page_offs, page_names, steps, check=True):
    """Stage times of the C5 chain and -- with the oracle as the checker -- `labels_equal_oracle`: the labels K11 produced
    equal oracle.cluster_images on the device's page matrix, and a seeded sample of page pairs equals the oracle's pair
    rule (wrc:199-226) on the kernel's own cosines.  (Every page pair is checked once in tests/test_gpu_pipeline.py.)"""
    import numpy as np

    S = c5["S"].cpu().numpy()
    out = {"pages": len(page_names), "regions_per_page": 128, "n_clusters": c5["k"],
           "stage_ms_per_step": {k: v * 1e3 / steps for k, v in c5["stage_s"].items()}}
    if not check:
        return out
    from multimodal_embeddings_amd.weighted_region_clustering import page_similarity_from_table
    from oracle import cluster as oclu
    from oracle import compare as ocmp

    t0 = time.perf_counter()
    want = oclu.cluster_images(S.copy(), list(page_names))
    out["labels_equal_oracle"] = bool(want is not None and want["labels"] == [int(v) for v in c5["labels"]] and want["n_clusters"] == c5["k"])
    out["oracle_cluster_s"] = time.perf_counter() - t0
    # sampled page pairs, raw values
    P = len(page_names)
    N = int(page_offs[-1])
    valid = np.ones(N, dtype=np.uint8)
    Sraw = page_similarity_from_table(table, area_pct, valid, page_offs, page_names, normalise=False, engine=eng).cpu().numpy()
    rng = np.random.default_rng(11)
    bad = checked = 0
    for i, j in rng.integers(0, P, (64, 2)):
        i, j = int(min(i, j)), int(max(i, j))
        if i == j:
            continue
        rows_i = np.arange(page_offs[i], page_offs[i + 1])
        rows_j = np.arange(page_offs[j], page_offs[j + 1])
        sims = eng.cosine(table[rows_i[:10].tolist()], table[rows_j.tolist()]).cpu().numpy()
        full = np.zeros((10, N), dtype=np.float32)
        full[:, rows_j] = sims

        class _Sim:
            def __getitem__(self, key):
                r, cand = key
                return full[int(r) - int(rows_i[0])][np.asarray(cand)]

        if page_names[i][:20] == page_names[j][:20]:
            want_ij = 0.0
        else:
            terms = ocmp.pair_terms(None, area_pct, rows_i, rows_j, len(rows_j), sim=_Sim())
            want_ij = float(np.sum(terms)) if terms else 0.0
        checked += 1
        if abs(Sraw[i, j] - want_ij) > 1e-13 * max(1.0, abs(want_ij)):
            bad += 1
    out["page_pairs_sampled"] = checked
    out["page_pairs_differing"] = bad
    out["labels_equal_oracle"] = bool(out["labels_equal_oracle"] and bad == 0)
    return out


if __name__ == "__main__":
    main()
