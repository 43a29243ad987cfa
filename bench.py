#!/usr/bin/env python3
"""Headline benchmark: region-crops/sec embedded + all-pairs cosine (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4]

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

Weights: seeded synthetic ViT-B/16 (no checkpoint can be fetched offline).

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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2_CROPS = 4096
C4_CROPS = 8192
C4_TABLE = 65536
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
    ap.add_argument("--config", choices=["auto", "c2", "c4"], default="auto", help="auto: c2 on one GPU, c4 on several")
    ap.add_argument("--crops", type=int, default=0, help="crops per GPU (overrides the config's 4096 / 8192)")
    ap.add_argument("--table-rows", type=int, default=-1, help="rows of the table the cosine block runs against "
                    "(-1: by config; rows beyond the gathered shards are seeded synthetic unit rows)")
    ap.add_argument("--chunk", type=int, default=0, help="crops per encoder pass (0 = library default)")
    ap.add_argument("--gemm-variant", type=int, default=0, help="0 auto, 1 128x128, 2 256x256 2-slot ring, 3 256x256 3-deep activation ring")
    ap.add_argument("--no-ln-fusion", action="store_true", help="separate LayerNorm kernel instead of folding it into the GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    written by tools/profile_record.py: which kernels' build, which passes, which clock).  None when absent."""
    try:
        with open(PROFILE_RECORD) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


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


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.timeout))

    import numpy as np
    import torch
    import torch.distributed as dist

    from multimodal_embeddings_amd import dist as mdist
    from multimodal_embeddings_amd._lib import Engine
    from multimodal_embeddings_amd.weights import make_vit_weights, synthetic_crops

    rank, world, local = mdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"error: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or run "
                  f"`python bench.py --gpus {args.gpus}` without a launcher (it starts its own ranks)", file=sys.stderr)
        sys.exit(2)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    config = args.config if args.config != "auto" else ("c2" if world == 1 else "c4")
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
    crops_host = synthetic_crops(n, seed=0, start=start)
    pix = torch.empty(n * 224 * 224 * 3 + 16, dtype=torch.uint8, device=dev)
    pix[: n * 224 * 224 * 3] = torch.from_numpy(crops_host.reshape(-1)).to(dev)
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

    def step(i=-1):
        eng.embed(pix, offs, hw, 0, out_f32=e32, out_bf16=e16)
        if world > 1:
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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
    elapsed = max(per_rank)
    steps = max(args.steps, 1)
    gather_ms = None
    if world > 1 and args.steps > 0:
        # device-side span of the collective on the launch stream (it includes waiting for the slowest rank's shard)
        gather_ms = max(mdist.all_gather_floats(sum(a.elapsed_time(b) for a, b in gather_ev[: args.steps]) / steps, dev))

    if rank == 0:
        ms_step = elapsed * 1e3 / steps
        total_crops = n * world * steps
        value = total_crops / elapsed
        gemm_ms, gemm_launches = prof["gemm"]
        gemm_ms_step = gemm_ms / steps
        ach = (FLOP_GEMM_PER_CROP * n) / (gemm_ms_step * 1e-3) / 1e12 if gemm_ms > 0 else None
        cos_ms = prof["cosine"][0] / steps if prof["cosine"][1] else None
        cos_bytes = float(n) * table_rows * 4 + float(table_rows) * 768 * 2 + float(n) * 768 * 2  # f32 block written + bf16 rows read
        cosine_hbm = {"achieved": cos_bytes / (cos_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cos_bytes / (cos_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                      "bytes": cos_bytes, "ms": cos_ms, "block": [n, table_rows]} if cos_ms else None
        pre_ms = prof["preprocess"][0] / steps if prof["preprocess"][1] else None
        preprocess_hbm = {"achieved": K1_BYTES_PER_CROP * n / (pre_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": K1_BYTES_PER_CROP * n / (pre_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), "bytes": float(K1_BYTES_PER_CROP) * n, "ms": pre_ms} if pre_ms else None
        rec = profile_record()
        traffic = rec.get("gemm_traffic_per_launch") if rec and rec.get("crops_per_gpu") == n else None
        roofline = {
            "kernel": "gemm_bf16_tn (K2/K4/K6/K7 launches of the ViT forward)",
            "bound": "mfma",
            "achieved": ach,
            "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": (ach / MFMA_BF16_PEAK_TFLOPS) if ach else None,
            "traffic": traffic,
            "traffic_unit": "bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes of this command; see from_profile)",
            "algorithmic_bytes_per_launch": (BYTES_GEMM_PER_CROP * n + BYTES_GEMM_WEIGHTS * (gemm_launches / steps / 49.0)) / (gemm_launches / steps) if gemm_launches else None,
            "launches_per_step": gemm_launches / steps,
            "avg_launch_ms": gemm_ms / gemm_launches if gemm_launches else None,
            "flop_per_launch_avg": FLOP_GEMM_PER_CROP * n / (gemm_launches / steps) if gemm_launches else None,
        }
        workload = (f"{config.upper()}: {n} synthetic 224x224x3 crops per GPU -> K1 patchify + ViT-B/16 bf16 forward + pool/L2 + "
                    f"[{n} x {table_rows}] cosine; seeded synthetic weights")
        if n_synth:
            workload += f"; {n_synth} table rows are seeded synthetic unit rows standing in for the shards of absent ranks"
        out = {
            "metric": baseline_metric_name(),
            "value": value,
            "unit": "region-crops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "crops_per_gpu": n,
                "table_rows": table_rows,
                "parallelism": f"dp{world} (crop shards, one all-gather of bf16 embeddings, backend {dist.get_backend()})" if world > 1 else "single GPU",
            },
            "ms_per_step_by_rank": [t * 1e3 / steps for t in per_rank],
            "allgather_ms": gather_ms,
            "allgather_host_ms": gather_host_s[0] * 1e3 / steps if world > 1 else None,
            "allgather_bytes": float(n) * world * 768 * 2 if world > 1 else None,
            "forward_mfma_frac": FLOP_FORWARD_PER_CROP * (n * steps / elapsed_local) / (MFMA_BF16_PEAK_TFLOPS * 1e12),
            "forward_mfma_frac_note": "forward FLOP x this rank's crops/s / 2.5 PFLOP/s nominal dense peak (no clock adjustment), measured in this run",
            "kernel_ms_per_step": {k: v[0] / steps for k, v in prof.items() if v[1]},
            "cosine_hbm": cosine_hbm,
            "preprocess_hbm": preprocess_hbm,
            "roofline": roofline,
            "from_profile": rec,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, ecpu = cpu_baseline(crops_host, weights)
            out["cpu_baseline"] = cb
            got = e32[: len(ecpu)].cpu().numpy()
            out["parity_max_1_minus_cos_vs_oracle"] = float(np.max(1.0 - np.sum(got * ecpu, axis=1)))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
