"""Host mirror of deprecated_package/embedder.py over the HIP engine.

Same surface as the reference's `MmE5MllamaEmbedder` (embedder.py:37-254):

    embedder = RegionEmbedder()                       # embedder.py:42
    vecs = embedder.get_image_embeddings(paths)       # embedder.py:141  list[list[float] | None]
    vec  = embedder.embed(region)                     # north_star: embed(region) -> vec

and the same error behaviour: `[]` for an empty list (:154-155), a `None` hole for any
item that fails to load, logged, never raised (:135-137), `[None]` for a failed single
query (:183-185).  What changes is the execution model: the reference loops one image per
forward on one thread per GPU (:104,:208); here the decoded crops of a call are packed
into one device buffer and go through K1 (resize/normalise/patchify) and the batched
bf16 MFMA ViT forward in `mme_embed` calls over bounded groups.  Under a launcher (LOCAL_RANK set)
one process drives one GPU and sharding a corpus across the 8 GPUs of a node is dist.py's job;
constructed plainly, as the reference's callers do, the embedder holds one context per visible
GPU (`gpu_count`, embedder.py:54-65) and fans a call out `i % n_devices` on one thread per
context (embedder.py:191-224).

There is no CPU path: constructing the embedder without libmme.so or without a GPU
raises `MmeError`.
"""
from __future__ import annotations

import logging
from concurrent.futures import ThreadPoolExecutor
import os

import numpy as np

from . import config
from ._lib import Engine, MmeError  # noqa: F401
from .weights import make_vit_weights

logger = logging.getLogger("multimodal_embeddings_amd")


def last_pooling(last_hidden_state, attention_mask, normalize=True):
    """embedder.py:17-34, tensor-for-tensor (torch, any device).

    Kept for API parity; inside the engine the same computation is kernel K8
    (`pool_ln_l2`), which gathers the pooled row straight from the residual stream.
    """
    import torch

    sequence_lengths = attention_mask.sum(dim=1) - 1
    batch_size = last_hidden_state.shape[0]
    reps = last_hidden_state[torch.arange(batch_size, device=last_hidden_state.device), sequence_lengths]
    if normalize:
        reps = torch.nn.functional.normalize(reps, p=2, dim=-1)
    return reps


def convert_to_rgb(image):
    """What the Mllama processor does to a non-RGB image before resizing (transformers
    image_processing_pil_mllama.py:196-211): composite over a WHITE background through RGBA, so transparent
    areas come out white (a plain `.convert("RGB")` would expose the colour stored under the alpha)."""
    from PIL import Image

    if image.mode == "RGB":
        return image
    rgba = image.convert("RGBA")
    background = Image.new("RGBA", rgba.size, (255, 255, 255))
    return Image.alpha_composite(background, rgba).convert("RGB")


def _load_rgb(item):
    """path | PIL.Image | uint8[h,w,3] -> contiguous uint8[h,w,3] (embedder.py:107-114)."""
    from PIL import Image

    if isinstance(item, np.ndarray):
        a = item
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"array crops must be uint8[h,w,3], got {a.dtype}{a.shape}")
        h, w = a.shape[:2]
        if h == 0 or w == 0:
            raise ValueError("empty image")
        if max(h, w) <= config.MAX_IMAGE_HEIGHT_AND_WIDTH:
            return np.ascontiguousarray(a)
        image = Image.fromarray(a)
    elif isinstance(item, (str, os.PathLike)):
        image = Image.open(item)
    else:
        image = item  # PIL image
    if image.size[0] > config.MAX_IMAGE_HEIGHT_AND_WIDTH or image.size[1] > config.MAX_IMAGE_HEIGHT_AND_WIDTH:
        scale = config.MAX_IMAGE_HEIGHT_AND_WIDTH / max(image.size)
        image = image.resize((int(image.size[0] * scale), int(image.size[1] * scale)), Image.LANCZOS)
    a = np.asarray(convert_to_rgb(image))
    if a.shape[0] == 0 or a.shape[1] == 0:
        raise ValueError("empty image")
    return np.ascontiguousarray(a)


class _Null:
    """Stream / event stand-in where there is no device (host-logic tests): every ordering call is a no-op."""

    def wait_event(self, *_):
        pass

    def record(self, *_):
        pass

    def synchronize(self):
        pass


class RegionEmbedder:
    """Drop-in for `MmE5MllamaEmbedder` on one MI355X."""

    GROUP_BYTES = 1 << 30  # packed pixels per mme_embed call (pinned staging + device copy stay bounded)
    TILE_GROUP_CROPS = 32  # crops per pass of the tile-ViT option (15 MB of f32 pixel values and ~230 MB of workspace each)

    def __init__(self, model_name=config.DEFAULT_MODEL_NAME, device=None, gpu_count=None, *, weights=None,
                 seed: int = 1, pool: str = "cls", chunk: int | None = None, engine: Engine | None = None, devices=None,
                 encoder: str = "vit_b16", geometry=None, prune_last_layer: bool | None = None):
        import torch

        self.torch = torch
        self.model_name = model_name
        if engine is not None:
            dev_list = [engine.device]
        elif devices is not None:  # explicit device indices (may repeat: several contexts on one GPU)
            dev_list = [int(d) for d in devices]
        elif isinstance(device, int):
            dev_list = [device]
        elif isinstance(device, str) and device not in ("cuda",):
            if not device.startswith("cuda"):
                raise MmeError(f"device={device!r}: this engine is HIP only (no CPU fallback)")
            dev_list = [int(device.split(":")[1]) if ":" in device else 0]
        elif "LOCAL_RANK" in os.environ:  # one process per GPU under torchrun / bench.py
            dev_list = [int(os.environ["LOCAL_RANK"])]
        else:  # embedder.py:54-65: gpu_count of the visible GPUs, all of them by default
            if not torch.cuda.is_available():
                raise MmeError("no GPU visible to torch; the embed/compare path is HIP only (no CPU fallback)")
            available = torch.cuda.device_count()
            n = available if gpu_count is None else max(1, min(int(gpu_count), available))
            logger.info(f"Using {n} of {available} available GPUs")
            dev_list = list(range(n))
        if not dev_list:
            raise MmeError("no device selected")
        self.gpu_count = len(dev_list)
        self.devices = [f"cuda:{d}" for d in dev_list]
        self.device = torch.device(self.devices[0])
        if encoder not in ("vit_b16", "mllama_tiles"):
            raise ValueError("encoder must be 'vit_b16' (BASELINE.json's re-scoped ViT-B/16) or 'mllama_tiles' (the checkpoint's own vision-tower geometry)")
        self.encoder = encoder
        if engine is not None:
            self.engines = [engine]
        else:
            if encoder == "mllama_tiles":
                from .weights import make_tile_vit_weights

                w = weights if weights is not None else (make_tile_vit_weights(seed + 1, geometry) if geometry else make_tile_vit_weights(seed + 1))
            else:
                w = weights if weights is not None else make_vit_weights(seed)
            self.engines = []
            for d in dev_list:  # one context (weights + workspace) per device, as embedder.py:73-82
                e = Engine(d)
                if encoder == "mllama_tiles":
                    e.load_tile_vit(w, geometry)
                else:
                    e.load_vit(w)
                self.engines.append(e)
        self.engine = self.engines[0]
        if chunk:
            for e in self.engines:
                e.set_chunk(chunk)
        if prune_last_layer is None:  # default: on for the contexts this object created, a caller's engine stays as it is
            prune_last_layer = engine is None
        if prune_last_layer and encoder == "vit_b16":
            # only the pooled token's row of the last layer is computed past its attention (mme_set_forward_pruning): callers of
            # this class only ever receive pooled vectors, and those are bit-identical, 6 % sooner.  The benchmark's headline
            # drives the Engine directly and times the whole forward; `prune_last_layer=False` restores that here.
            for e in self.engines:
                e.set_forward_pruning(True)
        if pool not in ("cls", "last"):
            raise ValueError("pool must be 'cls' or 'last'")
        self.pool_token = 0 if pool == "cls" else 196
        self._group_crops = 16 * config.BATCH_SIZE

    # -- device-resident API -------------------------------------------------------------------
    def pack(self, arrays, device=None):
        """list of uint8[h,w,3] -> (pix CUDA tensor, offs int64[n], hw int32[n,2])."""
        t = self.torch
        device = self.device if device is None else device
        n = len(arrays)
        hw = np.array([a.shape[:2] for a in arrays], dtype=np.int32).reshape(n, 2)
        sizes = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
        offs = np.zeros(n, dtype=np.int64)
        if n > 1:
            offs[1:] = np.cumsum((sizes[:-1] + 15) // 16 * 16)
        total = int(offs[-1] + sizes[-1]) + 16 if n else 16
        host = t.empty(total, dtype=t.uint8, pin_memory=t.cuda.is_available())
        hv = host.numpy()
        for a, o, s in zip(arrays, offs, sizes):
            hv[o : o + s] = a.reshape(-1)
        return host.to(device, non_blocking=True), offs, hw

    def embed_packed(self, pix, offs, hw, want_f32=True, want_bf16=True):
        """Packed crops already in HBM -> (f32 [n,768], bf16 [n,768]) CUDA tensors."""
        return self.engine.embed(pix, offs, hw, self.pool_token, want_f32=want_f32, want_bf16=want_bf16)

    def embed_uniform(self, crops):
        """uint8 CUDA tensor [n,224,224,3] (the synthetic C2/C4 workload) -> (f32, bf16)."""
        n = crops.shape[0]
        per = int(np.prod(crops.shape[1:]))
        offs = np.arange(n, dtype=np.int64) * per
        hw = np.tile(np.array([[crops.shape[1], crops.shape[2]]], dtype=np.int32), (n, 1))
        return self.engine.embed(crops.reshape(-1), offs, hw, self.pool_token)

    def process_images(self, images, tile=560, max_tiles=4):
        """What `self.processors[i](images=[image], return_tensors="pt")` hands the reference's model
        (embedder.py:117-121), for a batch of single-image samples, computed on the GPU (K1 multi-tile):
        {"pixel_values": f32 CUDA [B, 1, max_tiles, 3, tile, tile], "aspect_ratio_ids": int64 [B, 1],
         "aspect_ratio_mask": int64 [B, 1, max_tiles], "num_tiles": [[n], ...]} -- bit-exact with transformers'
        MllamaImageProcessorPil at the checkpoint geometry (tile 560, <= 4 tiles) when the engine's mean/std are
        the checkpoint's (the CLIP values by default)."""
        arrays = [_load_rgb(item) for item in images]
        pix, offs, hw = self.pack(arrays)
        pv, ids, mask, nt = self.engine.preprocess_tiles(pix, offs, hw, tile, max_tiles)
        return {"pixel_values": pv[:, None], "aspect_ratio_ids": ids[:, None], "aspect_ratio_mask": mask[:, None, :],
                "num_tiles": [[int(v)] for v in nt]}

    # -- reference surface ------------------------------------------------------------------------
    def _pipe_state(self, dev_idx):
        """Per-context staging of the host <-> device pipeline of `_embed_list`: two slots, each a pinned input buffer, its
        device twin and a pinned result buffer (grown on demand, kept), two copy streams (the H2D and D2H engines run
        side by side) and the events that order them against the compute stream."""
        st = self.__dict__.setdefault("_pipes", {})  # atomic: contexts are driven from one thread each, concurrently
        if dev_idx not in st:
            t = self.torch
            dev = t.device(self.devices[dev_idx])
            if dev.type != "cuda":  # the host-logic tests drive this pipeline with a stand-in engine and no device
                mk_stream = mk_event = _Null
            else:
                mk_stream, mk_event = (lambda: t.cuda.Stream(dev)), t.cuda.Event
            with self._on(dev):
                st[dev_idx] = {
                    "in_stream": mk_stream(), "out_stream": mk_stream(), "compute": None if dev.type == "cuda" else _Null(),
                    "pin": [None, None], "dev": [None, None], "out": [None, None], "pinned": dev.type == "cuda",
                    "used": [False, False],  # a slot's events mean something once it has been staged, in ANY earlier call
                    "ev_in": [mk_event(), mk_event()], "ev_done": [mk_event(), mk_event()], "ev_out": [mk_event(), mk_event()],
                }
        return st[dev_idx]

    def _on(self, dev, stream=None):
        """Context manager: `dev` current (and `stream` current on it); nothing to do without a device."""
        import contextlib

        t = self.torch
        if dev.type != "cuda":
            return contextlib.nullcontext()
        return t.cuda.stream(stream) if stream is not None else t.cuda.device(dev)

    def _stage_group(self, pipe, slot, arrays, device):
        """Pack decoded crops into the slot's pinned buffer (16-byte aligned, as `pack`) and start their H2D copy on the
        input stream.  Returns (device pixels, offs, hw).  The slot's previous occupants are out of the way: the caller
        acquired the slot after the consumer recorded `ev_done[slot]` (the pass that read the device twin)."""
        t = self.torch
        n = len(arrays)
        hw = np.array([a.shape[:2] for a in arrays], dtype=np.int32).reshape(n, 2)
        sizes = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
        offs = np.zeros(n, dtype=np.int64)
        if n > 1:
            offs[1:] = np.cumsum((sizes[:-1] + 15) // 16 * 16)
        total = int(offs[-1] + sizes[-1]) + 16
        first_use = not pipe["used"][slot]  # per pipe state, not per call: a call that left through an exception may have
        pipe["used"][slot] = True           # work in flight on the buffers the next call finds
        if pipe["pin"][slot] is None or pipe["pin"][slot].numel() < total:
            if not first_use:  # the buffers about to be dropped may still be read: by the last H2D, by the last pass
                pipe["ev_in"][slot].synchronize()
                pipe["ev_done"][slot].synchronize()
            cap = max(total, 1 << 20)
            pipe["pin"][slot] = t.empty(cap, dtype=t.uint8, pin_memory=pipe["pinned"])
            pipe["dev"][slot] = t.empty(cap, dtype=t.uint8, device=device)
        elif not first_use:
            pipe["ev_in"][slot].synchronize()  # the last H2D out of this pinned buffer has finished (long ago)
        hv = pipe["pin"][slot].numpy()

        def copy_range(lo, hi):  # large numpy copies release the GIL: a few threads fill the buffer side by side
            for k in range(lo, hi):
                o, sz = int(offs[k]), int(sizes[k])
                hv[o : o + sz] = arrays[k].reshape(-1)

        nbytes = int(sizes.sum())
        workers = min(4, os.cpu_count() or 1, max(1, nbytes >> 26))  # one thread per 64 MB, at most four
        if workers > 1:
            cuts = np.searchsorted(np.cumsum(sizes), np.linspace(0, nbytes, workers + 1)[1:-1]).tolist()
            bounds = [0] + [int(c) for c in cuts] + [n]
            with ThreadPoolExecutor(max_workers=workers) as pool:
                list(pool.map(lambda ab: copy_range(*ab), zip(bounds[:-1], bounds[1:])))
        else:
            copy_range(0, n)
        with self._on(device, pipe["in_stream"]):
            if not first_use:
                pipe["in_stream"].wait_event(pipe["ev_done"][slot])  # the pass that read the device twin has run
            pipe["dev"][slot][:total].copy_(pipe["pin"][slot][:total], non_blocking=True)
            pipe["ev_in"][slot].record(pipe["in_stream"])
        return pipe["dev"][slot][:total], offs, hw

    def _embed_list(self, dev_idx, items):
        """One context's share of a call: [(index, item)] -> [(index, list[float] | None)] (embedder.py:86-139).

        The items go through in bounded groups -- at most 16 x batch_size crops and GROUP_BYTES of packed pixels per
        `mme_embed` call -- each decoded, packed and embedded inside its own try/except: a failure (an unreadable
        file, an out-of-memory MmeError) voids only what it touched.

        The reference runs open -> processor -> forward -> `.cpu().tolist()` strictly in series per image
        (embedder.py:104-137).  Here the host side of group g + 1 (decode on a thread pool, packing into a pinned
        buffer, H2D on a copy stream) runs on a producer thread UNDER the device pass of group g, and the D2H of group
        g plus the float lists of group g - 1 behind it: two staging slots, three streams, events between them."""
        if getattr(self, "encoder", "vit_b16") == "mllama_tiles":
            return self._embed_list_serial(dev_idx, items)
        import queue
        import threading

        t = self.torch
        engine = self.engines[dev_idx]
        device = t.device(self.devices[dev_idx])
        pipe = self._pipe_state(dev_idx)
        results = []
        step = max(1, int(self._group_crops))
        ready = queue.Queue(maxsize=2)
        slot_free = [threading.Semaphore(1), threading.Semaphore(1)]
        stop = threading.Event()

        def load(pair):
            i, item = pair
            try:
                return i, _load_rgb(item)
            except Exception as e:  # embedder.py:135-137
                logger.error(f"Error processing image {item if isinstance(item, (str, os.PathLike)) else type(item)}: {e}")
                return i, None

        def produce():
            g = 0
            try:
                with self._on(device):
                    for g0 in range(0, len(items), step):
                        part = items[g0 : g0 + step]
                        # PNG decode is the slow part of a call and Pillow releases the GIL while decoding
                        if len(part) > 4 and not all(isinstance(it, np.ndarray) for _, it in part):
                            with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1, len(part))) as pool:
                                loaded = list(pool.map(load, part))
                        else:
                            loaded = [load(pr) for pr in part]
                        failed = [i for i, a in loaded if a is None]
                        groups, group, nbytes = [], [], 0
                        for i, a in loaded:
                            if a is None:
                                continue
                            if group and nbytes + a.nbytes > self.GROUP_BYTES:
                                groups.append(group)
                                group, nbytes = [], 0
                            group.append((i, a))
                            nbytes += a.nbytes
                        if group:
                            groups.append(group)
                        if failed:
                            ready.put(("failed", failed))
                        for group in groups:
                            slot = g & 1
                            while not slot_free[slot].acquire(timeout=0.1):
                                if stop.is_set():
                                    return
                            try:
                                staged = self._stage_group(pipe, slot, [a for _, a in group], device)
                                ready.put(("group", slot, [i for i, _ in group], staged))
                            except Exception as e:  # embedder.py:223-224
                                logger.error(f"Error in batch processing: {e}")
                                slot_free[slot].release()
                                ready.put(("failed", [i for i, _ in group]))
                                continue
                            g += 1
            except BaseException as e:  # never leave the consumer waiting
                logger.error(f"Error in batch processing: {e}")
            finally:
                ready.put(("end",))

        pending = None  # (slot, indices, n, device result kept alive) whose D2H is in flight

        def finalize(p):
            slot, idx, n, _keep = p
            pipe["ev_out"][slot].synchronize()
            if getattr(self, "_rows_as_array", False):
                rows = list(pipe["out"][slot][:n].numpy().copy())  # float32 row views (get_image_embeddings(as_array=True))
            else:
                # embedder.py:132 `.cpu().tolist()`.  In slices: one tolist() of a whole group holds the GIL for tens of
                # milliseconds, during which the producer thread cannot even start its next (GIL-free) buffer copy
                buf, rows = pipe["out"][slot], []
                for c0 in range(0, n, 64):
                    rows.extend(buf[c0 : min(n, c0 + 64)].tolist())
            results.extend(zip(idx, rows))

        producer = threading.Thread(target=produce, name=f"mme-stage-{dev_idx}", daemon=True)
        producer.start()
        try:
            with self._on(device):
                compute = pipe["compute"] or t.cuda.current_stream(device)
                while True:
                    msg = ready.get()
                    if msg[0] == "end":
                        break
                    if msg[0] == "failed":
                        results.extend((i, None) for i in msg[1])
                        continue
                    _, slot, idx, (pix, offs, hw) = msg
                    n = len(idx)
                    try:
                        compute.wait_event(pipe["ev_in"][slot])
                        e32, _ = engine.embed(pix, offs, hw, self.pool_token, want_bf16=False)
                        pipe["ev_done"][slot].record(compute)
                    except Exception as e:  # embedder.py:223-224
                        logger.error(f"Error in batch processing: {e}")
                        pipe["ev_done"][slot].record(compute)
                        slot_free[slot].release()
                        results.extend((i, None) for i in idx)
                        continue
                    slot_free[slot].release()  # the producer may refill the slot: its H2D waits for ev_done on the device
                    if pending is not None and pending[0] == slot:
                        finalize(pending)  # the result buffer of this slot is about to be reused
                        pending = None
                    if pipe["out"][slot] is None or pipe["out"][slot].shape[0] < n or pipe["out"][slot].shape[1] != e32.shape[1]:
                        pipe["out"][slot] = t.empty((max(n, 256), e32.shape[1]), dtype=t.float32, pin_memory=pipe["pinned"])
                    with self._on(device, pipe["out_stream"]):
                        pipe["out_stream"].wait_event(pipe["ev_done"][slot])
                        pipe["out"][slot][:n].copy_(e32, non_blocking=True)
                        pipe["ev_out"][slot].record(pipe["out_stream"])
                    if pending is not None:
                        finalize(pending)  # float lists of the previous group, under this group's device pass
                    pending = (slot, idx, n, e32)
                if pending is not None:
                    finalize(pending)
        except BaseException:
            if device.type == "cuda":  # leave nothing in flight on the staging buffers the next call will reuse
                t.cuda.synchronize(device)
            raise
        finally:
            stop.set()
            while producer.is_alive():  # drain so that a blocked put() returns
                try:
                    ready.get(timeout=0.05)
                except queue.Empty:
                    pass
            producer.join()
        return results

    def _embed_list_serial(self, dev_idx, items):
        """The unpipelined form (decode -> pack -> pass -> lists per group), used by the tile-ViT option."""
        engine = self.engines[dev_idx]
        device = self.torch.device(self.devices[dev_idx])
        results = []

        def load(pair):
            i, item = pair
            try:
                return i, _load_rgb(item)
            except Exception as e:  # embedder.py:135-137
                logger.error(f"Error processing image {item if isinstance(item, (str, os.PathLike)) else type(item)}: {e}")
                return i, None

        def run(group):
            if not group:
                return
            try:
                pix, offs, hw = self.pack([a for _, a in group], device)
                if getattr(self, "encoder", "vit_b16") == "mllama_tiles":  # processor (K1 multi-tile) + vision tower, class token of tile 0 (7680-d)
                    pv, ids, _, nt = engine.preprocess_tiles(pix, offs, hw, 560, 4)
                    _, e32, _ = engine.tile_vit_forward(pv, ids, nt, want_bf16=False)
                else:
                    e32, _ = engine.embed(pix, offs, hw, self.pool_token, want_bf16=False)
                rows = e32.cpu().tolist()
                results.extend((i, row) for (i, _), row in zip(group, rows))
            except Exception as e:  # embedder.py:223-224
                logger.error(f"Error in batch processing: {e}")
                results.extend((i, None) for i, _ in group)

        step = max(1, int(self._group_crops))
        if getattr(self, "encoder", "vit_b16") == "mllama_tiles":
            step = min(step, self.TILE_GROUP_CROPS)
        for g0 in range(0, len(items), step):
            part = items[g0 : g0 + step]
            if len(part) > 4:
                with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1, len(part))) as pool:
                    loaded = list(pool.map(load, part))
            else:
                loaded = [load(pr) for pr in part]
            group, nbytes = [], 0
            for i, a in loaded:
                if a is None:
                    results.append((i, None))
                    continue
                if group and nbytes + a.nbytes > self.GROUP_BYTES:
                    run(group)
                    group, nbytes = [], 0
                group.append((i, a))
                nbytes += a.nbytes
            run(group)
        return results

    def get_image_embeddings(self, image_paths, is_query=False, batch_size=config.BATCH_SIZE, *, as_array=False):
        """embedder.py:141-226: order-preserving list of float lists with None holes.

        `as_array=True` (an addition to the reference's signature) returns (float32 [n, D] ndarray, bool [n] mask of
        the rows that were embedded) instead: building n x D Python floats costs more host time than the device pass.

        Item i goes to context i % n_devices (:191-203) and every context works through its share on its own
        thread (:208-224); a single query image takes the first context without the pool (:158-185).
        `batch_size` keeps the reference's meaning "images per GPU per batch", except that a device pass here
        carries 16 of the reference's one-image forwards per unit of it (256 crops at the default 16)."""
        if not image_paths:
            return (np.zeros((0, 768), dtype=np.float32), np.zeros(0, dtype=bool)) if as_array else []
        self._rows_as_array = bool(as_array) and getattr(self, "encoder", "vit_b16") != "mllama_tiles"
        self._group_crops = 16 * max(1, int(batch_size))
        embeddings = [None] * len(image_paths)
        n_dev = len(self.engines)
        if n_dev == 1 or (len(image_paths) == 1 and is_query):
            shares = [list(enumerate(image_paths))]
        else:
            shares = [[(i, p) for i, p in enumerate(image_paths) if i % n_dev == d] for d in range(n_dev)]
        if len(shares) == 1:
            done = [self._embed_list(0, shares[0])]
        else:
            with ThreadPoolExecutor(max_workers=n_dev) as pool:
                done = list(pool.map(lambda d: self._embed_list(d, shares[d]), range(n_dev)))
        for part in done:
            for i, row in part:
                embeddings[i] = row
        if as_array:
            ok = np.array([row is not None for row in embeddings], dtype=bool)
            width = next((len(row) for row in embeddings if row is not None), 768)
            arr = np.zeros((len(embeddings), width), dtype=np.float32)
            for i, row in enumerate(embeddings):
                if row is not None:
                    arr[i] = row
            return arr, ok
        return embeddings

    def embed(self, region):
        """north_star `embed(region) -> vec`: one crop -> float32[768] (raises on failure)."""
        out = self.get_image_embeddings([region], is_query=True)
        if out[0] is None:
            raise MmeError("embed(region) failed; see log")
        return np.asarray(out[0], dtype=np.float32)

    def get_text_embeddings(self, text):
        """embedder.py:228-254 is adjacent to the hot path and not part of the metric."""
        raise NotImplementedError(
            "text embeddings need the mmE5 language tower, which BASELINE.json re-scopes away (ViT image encoder only)"
        )


# the reference's class name, so `from embedder import MmE5MllamaEmbedder` call sites port 1:1
MmE5MllamaEmbedder = RegionEmbedder
