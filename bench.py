#!/usr/bin/env python3
"""Headline benchmark: region-crops/sec embedded + all-pairs cosine (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over this rank's batch of synthetic crops, all inputs
already resident in HBM: K1 crop->normalise->patchify, ViT-B/16 bf16 MFMA forward, pooling
+ L2 (K2-K8), [N>1: one RCCL all-gather of the bf16 embedding shards], K9 cosine of the
local rows against all rows.  Workload C2 of SURVEY.md §8: 4096 synthetic 224x224x3 crops
per GPU (weak scaling; the cosine block is [4096, 4096*N]).  Weights: seeded synthetic
ViT-B/16 (no checkpoint can be fetched offline).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the bf16 MFMA GEMM,
gemm_bf16_tn_*): algorithmic GEMM FLOPs of a step / the GEMM kernels' summed duration,
measured with HIP events on the launch stream inside the timed region.  `cpu_baseline`
(N=1 only) times the oracle's reference-shaped per-crop loop on the host cores on a
bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CROPS_PER_GPU = 4096
FLOP_FORWARD_PER_CROP = 35_126_083_584  # SURVEY.md §8d (2*MAC, LN/softmax/GELU excluded)
FLOP_GEMM_PER_CROP = 231_211_008 + 12 * (697_171_968 + 232_390_656 + 2 * 929_562_624)  # K2,K4,K6,K7 launches
MFMA_BF16_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md (dense)
# algorithmic HBM bytes of the same launches per crop: every activation operand read once, every
# output written once (bf16), residual read once; weights (85.8 M bf16 per forward pass) added per pass
BYTES_GEMM_PER_CROP = 12 * 197 * 2 * ((768 + 2304) + 3 * 768 + (768 + 3072) + (3072 + 2 * 768)) + (196 + 197) * 768 * 2
BYTES_GEMM_WEIGHTS = 2 * (768 * 768 + 12 * (2304 * 768 + 768 * 768 + 2 * 3072 * 768))
TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", "round1_v10_gemm_traffic.json")  # tools/traffic_json.py, PMC passes of this bench


MFMA_UTIL_PROFILE = os.path.join(ROOT, "profiles", "round1_v9_mfma_util.json")  # from the SQ / GRBM PMC passes of this bench


def rocprof_mfma_util():
    """rocprof-reported matrix-pipe utilisation of the forward (busy cycles / available SIMD cycles at the clock the
    chip actually held), from the committed PMC passes; None when the record is absent."""
    try:
        with open(MFMA_UTIL_PROFILE) as fh:
            return float(json.load(fh)["forward_mfma_util"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def baseline_metric_name():
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, KeyError, ValueError):
        return "region-crops/sec embedded + all-pairs cosine, 224\u00d7224, 1/2/4/8 MI355X"


def measured_traffic():
    """HBM-side bytes per GEMM launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (separate runs of this same command; 2 x FETCH_SIZE + WRITE_SIZE, see profiles/README.md)."""
    try:
        with open(TRAFFIC_PROFILE) as fh:
            return float(json.load(fh)["gemm_traffic_per_launch"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def cpu_baseline(sample_crops: np.ndarray, weights, budget_s: float = 20.0):
    """Reference-shaped CPU path (embedder.py:104-137 loop: one crop per forward) via the oracle."""
    import torch

    from oracle import preprocess as opre
    from oracle import vit as ovit

    # the GPU box gives one job a 16-CPU share per GPU whatever os.cpu_count() says; more
    # threads than that only oversubscribe
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    embs = []
    t0 = time.perf_counter()
    done = 0
    for crop in sample_crops:
        patches = opre.preprocess_to_patches(crop)[None]
        embs.append(ovit.vit_embed(patches, weights, batch=1)[0])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    e = np.stack(embs)
    _ = e @ e.T
    dt = time.perf_counter() - t0
    # the same port at batch 16 (a fairer best case for the CPU than the reference's one-crop loop)
    t1 = time.perf_counter()
    nb = min(16, len(sample_crops))
    ovit.vit_embed(np.stack([opre.preprocess_to_patches(c) for c in sample_crops[:nb]]), weights, batch=nb)
    dt16 = time.perf_counter() - t1
    return {
        "value": done / dt,
        "value_batch16": nb / dt16,
        "unit": "region-crops/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{done} synthetic 224x224x3 crops, per-crop fp32 torch-CPU ViT-B/16 forward (batch 1, as embedder.py:104) + numpy cosine",
    }, e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--crops", type=int, default=CROPS_PER_GPU, help="crops per GPU (headline config: 4096)")
    ap.add_argument("--chunk", type=int, default=0, help="crops per encoder pass (0 = library default)")
    ap.add_argument("--gemm-variant", type=int, default=0, help="0 auto, 1 128x128, 2 256x256 2-slot ring, 3 256x256 3-deep activation ring")
    ap.add_argument("--no-ln-fusion", action="store_true", help="separate LayerNorm kernel instead of folding it into the GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from multimodal_embeddings_amd import dist as mdist
    from multimodal_embeddings_amd._lib import Engine
    from multimodal_embeddings_amd.weights import make_vit_weights, synthetic_crops

    rank, world, local = mdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    weights = make_vit_weights(seed=1)
    eng = Engine(local)
    eng.load_vit(weights)
    if args.chunk:
        eng.set_chunk(args.chunk)
    if args.gemm_variant:
        eng.set_gemm_variant(args.gemm_variant)
    if args.no_ln_fusion:
        eng.set_ln_fusion(False)

    n = args.crops
    start = rank * n
    crops_host = synthetic_crops(n, seed=0, start=start)
    pix = torch.empty(n * 224 * 224 * 3 + 16, dtype=torch.uint8, device=dev)
    pix[: n * 224 * 224 * 3] = torch.from_numpy(crops_host.reshape(-1)).to(dev)
    offs = np.arange(n, dtype=np.int64) * (224 * 224 * 3)
    hw = np.tile(np.array([[224, 224]], dtype=np.int32), (n, 1))
    e32 = torch.empty((n, 768), dtype=torch.float32, device=dev)
    e16 = torch.empty((n, 768), dtype=torch.bfloat16, device=dev)
    sim = torch.empty((n, n * world), dtype=torch.float32, device=dev)

    def step():
        eng.embed(pix, offs, hw, 0, out_f32=e32, out_bf16=e16)
        allv = mdist.all_gather_rows(e16)
        eng.cosine(e16, allv, out=sim)

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
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    elapsed = mdist.all_reduce_max_float(elapsed, dev)

    if rank == 0:
        steps = max(args.steps, 1)
        ms_step = elapsed * 1e3 / steps
        total_crops = n * world * steps
        value = total_crops / elapsed
        gemm_ms, gemm_launches = prof["gemm"]
        gemm_ms_step = gemm_ms / steps
        ach = (FLOP_GEMM_PER_CROP * n) / (gemm_ms_step * 1e-3) / 1e12 if gemm_ms > 0 else None
        cos_ms = prof["cosine"][0] / steps if prof["cosine"][1] else None
        cos_bytes = float(n) * n * world * 4 + float(n) * world * 768 * 2 + float(n) * 768 * 2  # f32 block written + bf16 rows read
        cosine_hbm = {"achieved": cos_bytes / (cos_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": cos_bytes / (cos_ms * 1e-3) / 8e12,
                      "bytes": cos_bytes} if cos_ms else None
        roofline = {
            "kernel": "gemm_bf16_tn (K2/K4/K6/K7 launches of the ViT forward)",
            "bound": "mfma",
            "achieved": ach,
            "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": (ach / MFMA_BF16_PEAK_TFLOPS) if ach else None,
            "traffic": measured_traffic() if n == CROPS_PER_GPU else None,
            "traffic_unit": "bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, profiles/round1_v10_gemm_traffic.json)",
            "algorithmic_bytes_per_launch": (BYTES_GEMM_PER_CROP * n + BYTES_GEMM_WEIGHTS * (gemm_launches / steps / 49.0)) / (gemm_launches / steps) if gemm_launches else None,
            "launches_per_step": gemm_launches / steps,
            "avg_launch_ms": gemm_ms / gemm_launches if gemm_launches else None,
            "flop_per_launch_avg": FLOP_GEMM_PER_CROP * n / (gemm_launches / steps) if gemm_launches else None,
        }
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
                "workload": f"C2: {n} synthetic 224x224x3 crops per GPU -> K1 patchify + ViT-B/16 bf16 forward + pool/L2 + "
                            f"[{n} x {n * world}] cosine; seeded synthetic weights",
                "crops_per_gpu": n,
                "parallelism": f"dp{world} (crop shards, one RCCL all-gather of bf16 embeddings)" if world > 1 else "single GPU",
            },
            "forward_mfma_frac": FLOP_FORWARD_PER_CROP * (n * steps / elapsed) / (MFMA_BF16_PEAK_TFLOPS * 1e12),
            "forward_mfma_util_rocprof": rocprof_mfma_util() if n == CROPS_PER_GPU else None,
            "kernel_ms_per_step": {k: v[0] / steps for k, v in prof.items() if v[1]},
            "cosine_hbm": cosine_hbm,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = crops_host[:64]
            cb, ecpu = cpu_baseline(sample, weights)
            out["cpu_baseline"] = cb
            got = e32[: len(ecpu)].cpu().numpy()
            out["parity_max_1_minus_cos_vs_oracle"] = float(np.max(1.0 - np.sum(got * ecpu, axis=1)))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
