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
bf16 MFMA ViT forward in a single `mme_embed` call.  One process drives one GPU (rank =
LOCAL_RANK); sharding a corpus across the 8 GPUs of a node is dist.py's job.

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


def _load_rgb(item):
    """path | PIL.Image | uint8[h,w,3] -> contiguous uint8[h,w,3] (embedder.py:107-114)."""
    from PIL import Image

    if isinstance(item, np.ndarray):
        a = item
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"array crops must be uint8[h,w,3], got {a.dtype}{a.shape}")
        h, w = a.shape[:2]
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
    a = np.asarray(image.convert("RGB"))
    if a.shape[0] == 0 or a.shape[1] == 0:
        raise ValueError("empty image")
    return np.ascontiguousarray(a)


class RegionEmbedder:
    """Drop-in for `MmE5MllamaEmbedder` on one MI355X."""

    def __init__(self, model_name=config.DEFAULT_MODEL_NAME, device=None, gpu_count=None, *, weights=None,
                 seed: int = 1, pool: str = "cls", chunk: int | None = None, engine: Engine | None = None):
        import torch

        self.torch = torch
        self.model_name = model_name
        if device is None or device == "cuda":
            dev_index = int(os.environ.get("LOCAL_RANK", "0"))
        elif isinstance(device, str):
            if not device.startswith("cuda"):
                raise MmeError(f"device={device!r}: this engine is HIP only (no CPU fallback)")
            dev_index = int(device.split(":")[1]) if ":" in device else 0
        else:
            dev_index = int(device)
        if gpu_count not in (None, 1):
            logger.info("gpu_count=%s ignored: one process drives one GPU; shard with dist.shard_range", gpu_count)
        self.gpu_count = 1
        self.devices = [f"cuda:{dev_index}"]
        self.device = torch.device(self.devices[0])
        self.engine = engine or Engine(dev_index)
        if engine is None:
            self.engine.load_vit(weights if weights is not None else make_vit_weights(seed))
        if chunk:
            self.engine.set_chunk(chunk)
        if pool not in ("cls", "last"):
            raise ValueError("pool must be 'cls' or 'last'")
        self.pool_token = 0 if pool == "cls" else 196

    # -- device-resident API -------------------------------------------------------------------
    def pack(self, arrays):
        """list of uint8[h,w,3] -> (pix CUDA tensor, offs int64[n], hw int32[n,2])."""
        t = self.torch
        n = len(arrays)
        hw = np.array([a.shape[:2] for a in arrays], dtype=np.int32).reshape(n, 2)
        sizes = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
        offs = np.zeros(n, dtype=np.int64)
        if n > 1:
            offs[1:] = np.cumsum((sizes[:-1] + 15) // 16 * 16)
        total = int(offs[-1] + sizes[-1]) + 16 if n else 16
        host = t.empty(total, dtype=t.uint8, pin_memory=True)
        hv = host.numpy()
        for a, o, s in zip(arrays, offs, sizes):
            hv[o : o + s] = a.reshape(-1)
        return host.to(self.device, non_blocking=True), offs, hw

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
    def get_image_embeddings(self, image_paths, is_query=False, batch_size=config.BATCH_SIZE):
        """embedder.py:141-226: order-preserving list of float lists with None holes."""
        if not image_paths:
            return []
        embeddings = [None] * len(image_paths)
        arrays, index = [], []

        def load(item):
            try:
                return _load_rgb(item)
            except Exception as e:  # embedder.py:135-137
                logger.error(f"Error processing image {item if isinstance(item, (str, os.PathLike)) else type(item)}: {e}")
                return None

        # PNG decode is the slow part of this call (the GPU needs ~3 ms for 48 crops) and Pillow releases the
        # GIL while decoding: decode on a small thread pool, keep the order
        if len(image_paths) > 4:
            with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1, len(image_paths))) as pool:
                loaded = list(pool.map(load, image_paths))
        else:
            loaded = [load(item) for item in image_paths]
        for i, a in enumerate(loaded):
            if a is not None:
                arrays.append(a)
                index.append(i)
        if not arrays:
            return embeddings
        try:
            pix, offs, hw = self.pack(arrays)
            e32, _ = self.embed_packed(pix, offs, hw, want_bf16=False)
            rows = e32.cpu().tolist()
        except MmeError as e:
            logger.error(f"Error in batch processing: {e}")  # embedder.py:223-224
            return embeddings
        for i, row in zip(index, rows):
            embeddings[i] = row
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
