"""Multi-GPU sharding: one process per GPU, one RCCL all-gather of embedding shards.

The reference's only parallelism is a replica per GPU inside one process with work
interleaved `i % n_devices` over a thread pool, results returned through Python lists
(deprecated_package/embedder.py:73-82,191-203,208-224); it has no collectives.  Here the
corpus is partitioned into contiguous blocks (page groups stay intact, which K10's page
segments rely on), each rank embeds its block, and the single exchange step of the
path is an all-gather of the `[n_local, 768]` bf16 shards (12.6 MB per GPU at N=65536)
so that every rank can compute its `[n_local, N]` row block of the cosine matrix.
xGMI is point to point (7 links per GPU): one bulk all-gather per step, no chatty
per-batch traffic.  backend "nccl" is RCCL on ROCm; the same code runs on gloo for the
CPU tests.
"""
from __future__ import annotations

import os

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition of n items: the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_pages(page_offs, rank: int, world: int) -> tuple[int, int]:
    """Partition whole pages so that region counts balance: returns a page range [p0, p1)."""
    page_offs = np.asarray(page_offs)
    P = len(page_offs) - 1
    total = int(page_offs[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(np.searchsorted(page_offs, target, side="left")))
    cuts.append(P)
    cuts = np.maximum.accumulate(np.clip(cuts, 0, P))
    return int(cuts[rank]), int(cuts[rank + 1])


_FORCED = False  # init_from_env(force=True): the collectives run on a process group of ONE rank too


def collectives_active() -> bool:
    """True when the collectives of this module go through torch.distributed: a process group of more than one rank,
    or one created with `force` (the world-size-1 rehearsal of the RCCL path: same calls, same buffers)."""
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCED)


def init_from_env(backend: str | None = None, force: bool = False):
    """torch.distributed init from RANK / WORLD_SIZE / MASTER_* (torchrun contract).

    `force` (or MME_FORCE_DIST=1) creates the process group at WORLD_SIZE = 1 as well and makes every collective of
    this module go through it: on the one-GPU box that is the only way the nccl (= RCCL) branch -- communicator
    creation on the device, `all_gather_into_tensor` straight into the table slice, the P x P all-reduce, barrier,
    destroy -- runs before the first real multi-GPU launch does."""
    global _FORCED
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = force or os.environ.get("MME_FORCE_DIST", "") not in ("", "0")
    if torch.cuda.is_available() and local >= torch.cuda.device_count():
        # more ranks than GPUs (a rehearsal on a smaller box): share devices round-robin; RCCL cannot
        # do that, so such a run needs MME_DIST_BACKEND=gloo
        local %= torch.cuda.device_count()
    if force:
        _FORCED = True
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("MME_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def all_gather_rows(local, counts=None, out=None):
    """Gather row blocks `[n_r, D]` from every rank into `[sum n_r, D]` on every rank.

    Equal shards use one `all_gather_into_tensor` (straight into `out` when the caller provides
    the `[sum n_r, D]` destination, e.g. a slice of a resident table); ragged shards are padded to
    the largest shard (counts = rows per rank, known from shard_range without communication).
    """
    import torch
    import torch.distributed as dist

    if not collectives_active():
        if out is not None:
            out.copy_(local)
            return out
        return local
    world = dist.get_world_size()
    n_local, d = local.shape
    if counts is None:
        counts = [n_local] * world
    counts = [int(c) for c in counts]
    assert counts[dist.get_rank()] == n_local, "counts disagree with the local shard"
    mx = max(counts)
    ragged = any(c != mx for c in counts)
    # gloo implements neither bf16 nor int16 all_gather: 16-bit payloads travel as bytes
    as_bytes = local.dtype == torch.bfloat16
    payload = local.contiguous().view(torch.uint8) if as_bytes else local
    db = payload.shape[1]
    if mx != n_local:
        pad = torch.zeros((mx - n_local, db), dtype=payload.dtype, device=payload.device)
        payload = torch.cat([payload, pad], dim=0)
    payload = payload.contiguous()
    if out is not None and not ragged:
        assert out.shape == (world * mx, d) and out.dtype == local.dtype and out.is_contiguous(), "out must be [sum n_r, D], contiguous"
        dist.all_gather_into_tensor(out.view(torch.uint8) if as_bytes else out, payload)
        return out
    got = torch.empty((world * mx, db), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(got, payload)
    if ragged:
        got = torch.cat([got[r * mx : r * mx + counts[r]] for r in range(world)], dim=0)
    got = got.view(torch.bfloat16) if as_bytes else got
    if out is not None:
        out.copy_(got)
        return out
    return got


def all_gather_floats(value: float, device=None) -> list:
    """One float per rank -> the list of all ranks' values on every rank ([value] without a process group)."""
    import torch
    import torch.distributed as dist

    if not collectives_active():
        return [float(value)]
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    got = torch.empty(dist.get_world_size(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(got, mine)
    return [float(v) for v in got.cpu()]


def all_reduce_max_float(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not collectives_active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(t):
    """In-place sum over ranks (no-op without an initialised process group); returns t."""
    import torch.distributed as dist

    if collectives_active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def normalise_page_matrix(S):
    """wrc:246-252 on a torch f64 [P, P] tensor: off-diagonal / max off-diagonal (if > 0), diagonal = 1."""
    import torch

    P = S.shape[0]
    eye = torch.eye(P, dtype=torch.bool, device=S.device)
    off = S.masked_fill(eye, float("-inf"))
    mx = off.max() if P > 1 else torch.tensor(0.0, dtype=S.dtype, device=S.device)
    out = torch.where(mx > 0, S / mx, S) if P > 1 else S.clone()
    return out.masked_fill(eye, 1.0)


def page_similarity_sharded(emb_all, area_percentage, valid, page_offs, image_names, *, rank=None, world=None, engine=None, **kwargs):
    """The page matrix with its P(P-1)/2 page pairs split evenly over the ranks: every rank
    holds all N embeddings (after `all_gather_rows`), computes the pairs of its `shard_range`, the partial
    matrices -- disjoint entries, zeros elsewhere -- are summed with one all-reduce of P*P f64 (2 MB at
    P = 512) and normalised identically on every rank.  Bit-identical to the single-GPU result."""
    import torch.distributed as dist

    from .weighted_region_clustering import page_similarity_from_table

    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    P = len(page_offs) - 1
    lo, hi = shard_range(P * (P - 1) // 2, rank, world)
    S = page_similarity_from_table(emb_all, area_percentage, valid, page_offs, image_names, normalise=False, engine=engine,
                                   pair_range=(lo, hi), **kwargs)
    return normalise_page_matrix(all_reduce_sum(S))


def neighbours_sharded(engine, emb_all, group=None, *, rank=None, world=None, gather=True, **kwargs):
    """K12 over the ranks: every rank holds all N unit rows (after `all_gather_rows`), ranks its own
    `shard_range(N)` block of query rows against all N and -- with gather=True -- the [N, top_n] index and
    similarity tables are assembled on every rank with two all-gathers (ragged shards allowed).
    Returns (idx int32 [rows, top_n], sim float32 [rows, top_n]); rows = N if gathered, else the shard."""
    import torch.distributed as dist

    ready = dist.is_available() and dist.is_initialized()
    world = (dist.get_world_size() if ready else 1) if world is None else world
    rank = (dist.get_rank() if ready else 0) if rank is None else rank
    n = emb_all.shape[0]
    lo, hi = shard_range(n, rank, world)
    idx, sim = engine.neighbours(emb_all, group, row0=lo, nrows=hi - lo, **kwargs)
    if not gather or not collectives_active():
        return idx, sim
    counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    return all_gather_rows(idx, counts), all_gather_rows(sim, counts)
