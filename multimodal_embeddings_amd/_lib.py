"""ctypes binding of libmme.so (include/mme.h).  Fails loudly: there is no CPU fallback.

Only this module touches the C ABI; the host mirrors of the reference interface
(embedder.py, cross_compare.py, weighted_region_clustering.py) go through `Engine`.
torch is used for device memory and streams only (data_ptr() into the ABI).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library is the in-tree libmme.so.  Measurement tools load libmme_diag.so (build.py --diag: the only build
# that reads the experiment switches of DESIGN.md 4.5) by naming it in MME_LIB_PATH AND opting in with
# MME_ALLOW_LIB_OVERRIDE=1 (tools/_diag.py sets both): one stray variable in a user's environment must not be able to
# swap the library -- load_library refuses the override without the opt-in, and says so on stderr when a diagnostic
# build is what got loaded.
DEFAULT_LIB_PATH = os.path.join(_HERE, "libmme.so")
LIB_PATH = os.environ.get("MME_LIB_PATH") or DEFAULT_LIB_PATH
DIAG_LIB_PATH = os.path.join(_HERE, "libmme_diag.so")

NUM_KERNEL_CLASSES = 10
ABI_VERSION = 2  # include/mme.h MME_ABI_VERSION this binding was written against
KERNEL_CLASSES = ("preprocess", "gemm", "layernorm", "attention", "pool", "cosine", "page_reduce", "cluster", "neighbours", "allgather")


class MmeError(RuntimeError):
    pass


class _Layer(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in (
        "ln1_g", "ln1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b",
        "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class _Weights(C.Structure):
    _fields_ = [
        ("image_size", C.c_int32), ("patch_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32),
        ("heads", C.c_int32), ("mlp", C.c_int32), ("ln_eps", C.c_float),
        ("cls_token", C.POINTER(C.c_float)), ("pos_emb", C.POINTER(C.c_float)),
        ("patch_w", C.POINTER(C.c_float)), ("patch_b", C.POINTER(C.c_float)),
        ("lnf_g", C.POINTER(C.c_float)), ("lnf_b", C.POINTER(C.c_float)),
        ("layer", C.POINTER(_Layer)),
    ]


class _TileLayer(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in ("ln1_g", "ln1_b", "q_w", "k_w", "v_w", "o_w", "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")] + [
        ("gate_attn", C.c_float), ("gate_ffn", C.c_float), ("gated", C.c_int32)]


class _TileWeights(C.Structure):
    _fields_ = [
        ("image_size", C.c_int32), ("patch_size", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("mlp", C.c_int32),
        ("max_tiles", C.c_int32), ("aspect_ratios", C.c_int32), ("layers", C.c_int32), ("global_layers", C.c_int32),
        ("n_intermediate", C.c_int32), ("intermediate", C.c_int32 * 8), ("intermediate_save_point", C.c_int32), ("norm_eps", C.c_float),
        ("pos_gate", C.c_float), ("pre_gate", C.c_float), ("post_gate", C.c_float),
        ("class_embedding", C.POINTER(C.c_float)), ("patch_w", C.POINTER(C.c_float)), ("pos_emb", C.POINTER(C.c_float)),
        ("tile_pos_emb", C.POINTER(C.c_float)), ("pre_emb", C.POINTER(C.c_float)), ("post_emb", C.POINTER(C.c_float)),
        ("ln_pre_g", C.POINTER(C.c_float)), ("ln_pre_b", C.POINTER(C.c_float)), ("ln_post_g", C.POINTER(C.c_float)), ("ln_post_b", C.POINTER(C.c_float)),
        ("layer", C.POINTER(_TileLayer)),
    ]


EXPORTS = {
    "mme_abi_version": (C.c_int, []),
    "mme_is_diag_build": (C.c_int, []),
    "mme_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mme_destroy": (None, [C.c_void_p]),
    "mme_last_error": (C.c_char_p, [C.c_void_p]),
    "mme_load_vit": (C.c_int, [C.c_void_p, C.POINTER(_Weights)]),
    "mme_set_normalisation": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mme_set_chunk": (C.c_int, [C.c_void_p, C.c_int]),
    "mme_set_gemm_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "mme_set_ln_fusion": (C.c_int, [C.c_void_p, C.c_int]),
    "mme_set_attention_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "mme_attention_redone": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mme_set_forward_pruning": (C.c_int, [C.c_void_p, C.c_int]),
    "mme_set_tile_order": (C.c_int, [C.c_void_p, C.c_int]),
    "mme_preprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mme_vit_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mme_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mme_normalise_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mme_cosine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mme_cosine_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mme_page_similarity": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mme_page_similarity_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_int, C.c_int, C.c_double, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "mme_cluster_pages": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mme_preprocess_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "mme_crop_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mme_nms_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mme_neighbours": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mme_set_neighbour_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "mme_gemm_bench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "mme_gemm_stamps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mme_load_tile_vit": (C.c_int, [C.c_void_p, C.POINTER(_TileWeights)]),
    "mme_tile_vit_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mme_comm_unique_id": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mme_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mme_comm_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mme_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "mme_attention_stamps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "mme_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mme_profile_reset": (C.c_int, [C.c_void_p]),
    "mme_profile_read_sync": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def load_library(path: str | None = None):
    """dlopen libmme.so and type every export of include/mme.h (no GPU needed for this)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if path is None and os.path.abspath(p) != os.path.abspath(DEFAULT_LIB_PATH) and os.environ.get("MME_ALLOW_LIB_OVERRIDE") != "1":
        raise MmeError(f"MME_LIB_PATH={p} names another library than the in-tree libmme.so; that is a measurement-tool switch "
                       "(tools/_diag.py) and needs MME_ALLOW_LIB_OVERRIDE=1 beside it.  Unset MME_LIB_PATH to run the product library.")
    # torch ships its own libamdhip64; libmme.so must bind to THAT runtime (one HIP runtime per
    # process), so torch is always loaded first.  Loading libmme.so first makes the second
    # runtime report "no ROCm-capable device".
    import torch  # noqa: F401

    if not os.path.exists(p):
        raise MmeError(
            f"{p} is missing: build it with `python -m multimodal_embeddings_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback for the embed/compare path."
        )
    lib = C.CDLL(p)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mme_abi_version() != ABI_VERSION:
        raise MmeError(f"libmme ABI version {lib.mme_abi_version()} != {ABI_VERSION} (include/mme.h MME_ABI_VERSION): rebuild the library "
                       "(python -m multimodal_embeddings_amd.build --force) or update the binding")
    if lib.mme_is_diag_build():
        import sys

        print(f"libmme: DIAGNOSTIC build loaded ({p}): it reads the MME_* experiment switches from the environment, some of which "
              "produce wrong results on purpose; never use it for production embeddings", file=sys.stderr, flush=True)
    if path is None:
        _lib = lib
    return lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Engine:
    """One mme_ctx bound to one GPU.  Methods take torch CUDA tensors / numpy control arrays."""

    def __init__(self, device: int = 0):
        import torch

        self.lib = load_library()
        if not torch.cuda.is_available():
            raise MmeError("no GPU visible to torch; the embed/compare path is HIP only (no CPU fallback)")
        self.torch = torch
        self.device = int(device)
        h = C.c_void_p()
        rc = self.lib.mme_create(self.device, C.byref(h))
        if rc != 0:
            raise MmeError(f"mme_create({device}) failed ({rc}): {self.lib.mme_last_error(None).decode()}")
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.mme_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise MmeError(f"{what} failed ({rc}): {self.lib.mme_last_error(self.h).decode()}")

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # ---- weights ---------------------------------------------------------------------------------
    def load_vit(self, w: dict, eps: float = 1e-12):
        def arr(name):
            a = np.ascontiguousarray(w[name], dtype=np.float32)
            self._keep.append(a)
            return _fp(a)

        layers = (_Layer * 12)()
        for i in range(12):
            p = f"layers.{i}."
            L = layers[i]
            L.ln1_g, L.ln1_b = arr(p + "layernorm_before.weight"), arr(p + "layernorm_before.bias")
            L.q_w, L.q_b = arr(p + "attention.q_proj.weight"), arr(p + "attention.q_proj.bias")
            L.k_w, L.k_b = arr(p + "attention.k_proj.weight"), arr(p + "attention.k_proj.bias")
            L.v_w, L.v_b = arr(p + "attention.v_proj.weight"), arr(p + "attention.v_proj.bias")
            L.o_w, L.o_b = arr(p + "attention.o_proj.weight"), arr(p + "attention.o_proj.bias")
            L.ln2_g, L.ln2_b = arr(p + "layernorm_after.weight"), arr(p + "layernorm_after.bias")
            L.fc1_w, L.fc1_b = arr(p + "mlp.fc1.weight"), arr(p + "mlp.fc1.bias")
            L.fc2_w, L.fc2_b = arr(p + "mlp.fc2.weight"), arr(p + "mlp.fc2.bias")
        W = _Weights(224, 16, 768, 12, 12, 3072, float(eps))
        W.cls_token = arr("embeddings.cls_token")
        W.pos_emb = arr("embeddings.position_embeddings")
        W.patch_w = arr("embeddings.patch_embeddings.projection.weight")
        W.patch_b = arr("embeddings.patch_embeddings.projection.bias")
        W.lnf_g, W.lnf_b = arr("layernorm.weight"), arr("layernorm.bias")
        W.layer = layers
        self._check(self.lib.mme_load_vit(self.h, C.byref(W)), "mme_load_vit")
        self._keep.clear()

    def load_tile_vit(self, w: dict, geom=None):
        """Hugging Face `MllamaVisionModel` state dict (f32 arrays) -> the tile-ViT encoder of this context."""
        from .weights import TILE_VIT

        geom = geom or TILE_VIT
        keep = []

        def arr(name):
            a = np.ascontiguousarray(w[name], dtype=np.float32)
            keep.append(a)
            return _fp(a)

        L = geom.num_layers + geom.num_global_layers
        layers = (_TileLayer * L)()
        for i in range(L):
            gated = i >= geom.num_layers
            p = f"global_transformer.layers.{i - geom.num_layers}." if gated else f"transformer.layers.{i}."
            X = layers[i]
            X.ln1_g, X.ln1_b = arr(p + "input_layernorm.weight"), arr(p + "input_layernorm.bias")
            X.q_w, X.k_w, X.v_w, X.o_w = (arr(p + f"self_attn.{n}_proj.weight") for n in "qkvo")
            X.ln2_g, X.ln2_b = arr(p + "post_attention_layernorm.weight"), arr(p + "post_attention_layernorm.bias")
            X.fc1_w, X.fc1_b, X.fc2_w, X.fc2_b = arr(p + "mlp.fc1.weight"), arr(p + "mlp.fc1.bias"), arr(p + "mlp.fc2.weight"), arr(p + "mlp.fc2.bias")
            X.gated = int(gated)
            X.gate_attn = float(w[p + "gate_attn"][0]) if gated else 0.0
            X.gate_ffn = float(w[p + "gate_ffn"][0]) if gated else 0.0
        W = _TileWeights()
        W.image_size, W.patch_size, W.hidden, W.heads, W.mlp = geom.image_size, geom.patch_size, geom.hidden_size, geom.num_heads, geom.intermediate_size
        W.max_tiles, W.aspect_ratios, W.layers, W.global_layers = geom.max_num_tiles, geom.max_aspect_ratio_id + 1, geom.num_layers, geom.num_global_layers
        W.n_intermediate = len(geom.intermediate_layers)
        for k, v in enumerate(geom.intermediate_layers):
            W.intermediate[k] = int(v)
        W.intermediate_save_point = {"after": 0, "before": 1}[geom.intermediate_save_point]
        W.norm_eps = float(geom.norm_eps)
        W.pos_gate = float(w["gated_positional_embedding.gate"][0])
        W.pre_gate = float(w["pre_tile_positional_embedding.gate"][0])
        W.post_gate = float(w["post_tile_positional_embedding.gate"][0])
        W.class_embedding, W.patch_w = arr("class_embedding"), arr("patch_embedding.weight")
        W.pos_emb, W.tile_pos_emb = arr("gated_positional_embedding.embedding"), arr("gated_positional_embedding.tile_embedding.weight")
        W.pre_emb, W.post_emb = arr("pre_tile_positional_embedding.embedding.weight"), arr("post_tile_positional_embedding.embedding.weight")
        W.ln_pre_g, W.ln_pre_b = arr("layernorm_pre.weight"), arr("layernorm_pre.bias")
        W.ln_post_g, W.ln_post_b = arr("layernorm_post.weight"), arr("layernorm_post.bias")
        W.layer = layers
        self._check(self.lib.mme_load_tile_vit(self.h, C.byref(W)), "mme_load_tile_vit")
        self.tile_features = geom.output_dim

    def tile_vit_forward(self, pixel_values, aspect_ratio_ids, num_tiles, want_hidden=False, want_f32=True, want_bf16=True):
        """pixel_values f32 CUDA [n, 4, 3, 560, 560] (+ ids / tile counts, as `preprocess_tiles` returns them) ->
        (hidden f32 [n, 4, 1601, F] | None, emb f32 [n, F] | None, emb bf16 [n, F] | None)."""
        t = self.torch
        pv = pixel_values.contiguous()
        n, F = pv.shape[0], self.tile_features
        if pv.dtype != t.float32 or tuple(pv.shape[1:]) != (4, 3, 560, 560):
            raise MmeError("tile_vit_forward: pixel_values must be f32 [n, 4, 3, 560, 560]")
        ids = np.ascontiguousarray(np.asarray(aspect_ratio_ids).reshape(-1), dtype=np.int32)
        nt = np.ascontiguousarray(np.asarray(num_tiles).reshape(-1), dtype=np.int32)
        hidden = t.empty((n, 4, 1601, F), dtype=t.float32, device=pv.device) if want_hidden else None
        e32 = t.empty((n, F), dtype=t.float32, device=pv.device) if want_f32 else None
        e16 = t.empty((n, F), dtype=t.bfloat16, device=pv.device) if want_bf16 else None
        self._check(self.lib.mme_tile_vit_forward(self.h, pv.data_ptr(), ids.ctypes.data, nt.ctypes.data, n,
                                                  hidden.data_ptr() if want_hidden else None, e32.data_ptr() if want_f32 else None,
                                                  e16.data_ptr() if want_bf16 else None, self._stream()), "mme_tile_vit_forward")
        return hidden, e32, e16

    def set_normalisation(self, mean, std):
        m = (C.c_float * 3)(*mean)
        s = (C.c_float * 3)(*std)
        self._check(self.lib.mme_set_normalisation(self.h, m, s), "mme_set_normalisation")

    def set_gemm_variant(self, variant: int):
        self._check(self.lib.mme_set_gemm_variant(self.h, int(variant)), "mme_set_gemm_variant")

    def set_ln_fusion(self, mode):
        """0 / False: LayerNorm kernel; 1: folded + statistics pass over x; 2 / True: folded + partial sums from the producing GEMM."""
        m = 2 if mode is True else int(mode)
        self._check(self.lib.mme_set_ln_fusion(self.h, m), "mme_set_ln_fusion")

    def set_attention_mode(self, mode):
        """1 / "fast" (default): guarded fast softmax; 0 / "exact": row maximum first (see mme.h)."""
        m = {"exact": 0, "fast": 1, "fast_forced_redo": 2}.get(mode, mode)
        self._check(self.lib.mme_set_attention_mode(self.h, int(m)), "mme_set_attention_mode")

    def set_tile_order(self, mode: int):
        """0 upwards, 1 zig-zag between consecutive kernels (default), 2 attention downwards only; bit-identical results."""
        self._check(self.lib.mme_set_tile_order(self.h, int(mode)), "mme_set_tile_order")

    def set_forward_pruning(self, on: bool):
        """Skip what nothing reads in the LAST layer (only the pooled token's row is computed after its attention):
        bit-identical embeddings, 6 % less work; off by default (see mme.h)."""
        self._check(self.lib.mme_set_forward_pruning(self.h, int(bool(on))), "mme_set_forward_pruning")

    def attention_redone(self):
        """Layers of the LAST encoder pass whose attention launch raised the fast form's guard (list of 12 ints)."""
        flags = (C.c_int32 * 12)()
        self._check(self.lib.mme_attention_redone(self.h, flags), "mme_attention_redone")
        return list(flags)

    def set_chunk(self, crops: int):
        self._check(self.lib.mme_set_chunk(self.h, int(crops)), "mme_set_chunk")

    # ---- hot path ----------------------------------------------------------------------------------
    def _crop_tables(self, offs, hw):
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        hw = np.ascontiguousarray(hw, dtype=np.int32).reshape(-1, 2)
        if len(offs) != len(hw):
            raise ValueError("offs and hw disagree")
        return offs, hw

    def crop_boxes(self, page, boxes, out=None, base: int = 0):
        """page: uint8 CUDA tensor [H, W, 3]; boxes: int array [n, 4] (x0, y0, x1, y1), already int()-truncated.

        Returns (pix uint8 CUDA tensor, offs int64[n], hw int32[n, 2]) ready for `preprocess` / `embed`.  With `out` (a
        uint8 CUDA buffer) the crops are packed into it from byte `base` (a multiple of 16) on and `offs` are offsets
        into `out`: the boxes of several pages fill ONE packed buffer (RegionProcessor.process_regions)."""
        t = self.torch
        if page.dtype != t.uint8 or page.dim() != 3 or page.shape[2] != 3 or not page.is_contiguous():
            raise MmeError("crop_boxes: page must be a contiguous uint8 [H, W, 3] tensor")
        b = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 4))
        n = len(b)
        hw = np.ascontiguousarray(np.stack([b[:, 3] - b[:, 1], b[:, 2] - b[:, 0]], axis=1).astype(np.int32)) if n else np.zeros((0, 2), np.int32)
        if n and (hw.min() <= 0 or hw.max() > 8000):
            raise MmeError("crop_boxes: every box must be 1..8000 pixels wide and high after int() truncation")
        size = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
        offs = np.zeros(n, dtype=np.int64)
        if n > 1:
            offs[1:] = np.cumsum((size[:-1] + 15) // 16 * 16)
        total = int(offs[-1] + size[-1]) if n else 0
        if out is not None:
            if out.dtype != t.uint8 or not out.is_contiguous() or out.device != page.device or base % 16 or base < 0 or base + total + 16 > out.numel():
                raise MmeError("crop_boxes: `out` must be a contiguous uint8 buffer on the page's device with room for the crops (+16 B) from a 16-byte aligned `base`")
            pix, offs = out, offs + int(base)
        else:
            pix = t.empty(total + 16, dtype=t.uint8, device=page.device)
        self._check(self.lib.mme_crop_boxes(self.h, page.data_ptr(), int(page.shape[0]), int(page.shape[1]), b.ctypes.data, n,
                                            pix.data_ptr(), offs.ctypes.data, self._stream()), "mme_crop_boxes")
        return pix, offs, hw

    def nms_boxes(self, boxes, scores, classes, page_offs, iou_threshold=0.5):
        """K13 (3_combine_grids.py:80-137) over many pages: host arrays in, list of kept page-local index arrays out
        (the reference's output order)."""
        boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.float64).reshape(-1, 4))
        scores = np.ascontiguousarray(scores, dtype=np.float64).reshape(-1)
        classes = np.ascontiguousarray(classes, dtype=np.int32).reshape(-1)
        page_offs = np.ascontiguousarray(page_offs, dtype=np.int32).reshape(-1)
        pages = len(page_offs) - 1
        n = len(scores)
        if pages < 0 or len(boxes) != n or len(classes) != n or (pages >= 0 and int(page_offs[-1]) != n):
            raise MmeError("nms_boxes: boxes / scores / classes / page_offs disagree")
        keep = np.full(n, -1, dtype=np.int32)
        count = np.zeros(max(pages, 0), dtype=np.int32)
        self._check(self.lib.mme_nms_boxes(self.h, boxes.ctypes.data, scores.ctypes.data, classes.ctypes.data, page_offs.ctypes.data,
                                           pages, float(iou_threshold), keep.ctypes.data, count.ctypes.data, self._stream()), "mme_nms_boxes")
        return [keep[page_offs[p] : page_offs[p] + count[p]].copy() for p in range(pages)]

    def preprocess(self, pix, offs, hw):
        """pix: uint8 CUDA tensor (concatenated HWC crops, >=16 spare bytes at the end)."""
        t = self.torch
        offs, hw = self._crop_tables(offs, hw)
        n = len(offs)
        patches = t.empty((n * 196, 768), dtype=t.bfloat16, device=pix.device)
        self._check(self.lib.mme_preprocess(self.h, pix.data_ptr(), offs.ctypes.data, hw.ctypes.data, n, patches.data_ptr(), self._stream()), "mme_preprocess")
        return patches

    def preprocess_tiles(self, pix, offs, hw, tile=560, max_tiles=4):
        """Mllama multi-tile preprocessing of packed crops -> (pixel_values f32 CUDA [n, max_tiles, 3, tile, tile],
        aspect_ratio_ids int64[n], aspect_ratio_mask int64[n, max_tiles], num_tiles list[int])."""
        t = self.torch
        offs, hw = self._crop_tables(offs, hw)
        n = len(offs)
        out = t.empty((n, max_tiles, 3, tile, tile), dtype=t.float32, device=pix.device)
        ids = np.zeros(n, dtype=np.int32)
        nt = np.zeros(n, dtype=np.int32)
        self._check(self.lib.mme_preprocess_tiles(self.h, pix.data_ptr(), offs.ctypes.data, hw.ctypes.data, n, int(tile), int(max_tiles),
                                                  out.data_ptr(), ids.ctypes.data, nt.ctypes.data, self._stream()), "mme_preprocess_tiles")
        mask = (np.arange(max_tiles)[None, :] < nt[:, None]).astype(np.int64)
        return out, ids.astype(np.int64), mask, nt.tolist()

    def vit_forward(self, patches, pool_token: int = 0, want_f32: bool = True, want_bf16: bool = True):
        t = self.torch
        n = patches.shape[0] // 196
        e32 = t.empty((n, 768), dtype=t.float32, device=patches.device) if want_f32 else None
        e16 = t.empty((n, 768), dtype=t.bfloat16, device=patches.device) if want_bf16 else None
        self._check(self.lib.mme_vit_forward(self.h, patches.data_ptr(), n, int(pool_token), e32.data_ptr() if want_f32 else None,
                                             e16.data_ptr() if want_bf16 else None, self._stream()), "mme_vit_forward")
        return e32, e16

    def embed(self, pix, offs, hw, pool_token: int = 0, out_f32=None, out_bf16=None, want_f32: bool = True, want_bf16: bool = True):
        t = self.torch
        offs, hw = self._crop_tables(offs, hw)
        n = len(offs)
        e32 = out_f32 if out_f32 is not None else (t.empty((n, 768), dtype=t.float32, device=pix.device) if want_f32 else None)
        e16 = out_bf16 if out_bf16 is not None else (t.empty((n, 768), dtype=t.bfloat16, device=pix.device) if want_bf16 else None)
        self._check(self.lib.mme_embed(self.h, pix.data_ptr(), offs.ctypes.data, hw.ctypes.data, n, int(pool_token),
                                       e32.data_ptr() if e32 is not None else None, e16.data_ptr() if e16 is not None else None,
                                       self._stream()), "mme_embed")
        return e32, e16

    def normalise_rows(self, x):
        """f32 CUDA tensor [n,d] -> L2-normalised bf16 [n,d]."""
        t = self.torch
        x = x.contiguous()
        y = t.empty(x.shape, dtype=t.bfloat16, device=x.device)
        self._check(self.lib.mme_normalise_rows(self.h, x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), self._stream()), "mme_normalise_rows")
        return y

    def cosine(self, a, b=None, out=None):
        """a [m,d], b [n,d] bf16 CUDA tensors of L2-normalised rows -> f32 [m,n]."""
        t = self.torch
        b = a if b is None else b
        m, d = a.shape
        n = b.shape[0]
        if out is None:
            out = t.empty((m, n), dtype=t.float32, device=a.device)
        assert a.dtype == t.bfloat16 and b.dtype == t.bfloat16 and a.is_contiguous() and b.is_contiguous()
        self._check(self.lib.mme_cosine(self.h, a.data_ptr(), m, b.data_ptr(), n, d, out.data_ptr(), out.stride(0), self._stream()), "mme_cosine")
        return out

    def cosine_bf16(self, a, b=None, out=None):
        """as `cosine`, S rounded to bf16 (mme_cosine_bf16): half the bytes; bit-equal to `cosine(...).to(bfloat16)`."""
        t = self.torch
        b = a if b is None else b
        m, d = a.shape
        n = b.shape[0]
        if out is None:
            out = t.empty((m, (n + 7) // 8 * 8), dtype=t.bfloat16, device=a.device)[:, :n]
        assert a.dtype == t.bfloat16 and b.dtype == t.bfloat16 and a.is_contiguous() and b.is_contiguous() and out.dtype == t.bfloat16
        self._check(self.lib.mme_cosine_bf16(self.h, a.data_ptr(), m, b.data_ptr(), n, d, out.data_ptr(), out.stride(0), self._stream()), "mme_cosine_bf16")
        return out

    def page_similarity(self, emb, area_pct, valid, page_offs, skip=None, *, max_query=10, top_k=10, max_dist=0.9, metric=0, normalise=True,
                        pair_range=None):
        """pair_range=(lo, hi): only those upper-triangle pair ranks, raw and zero elsewhere (one rank's shard)."""
        t = self.torch
        N, d = emb.shape
        page_offs = np.ascontiguousarray(page_offs, dtype=np.int32)
        P = len(page_offs) - 1
        S = t.empty((P, P), dtype=t.float64, device=emb.device)
        if pair_range is not None:
            self._check(self.lib.mme_page_similarity_pairs(self.h, emb.data_ptr(), N, d, area_pct.data_ptr(), valid.data_ptr(), page_offs.ctypes.data,
                                                           P, skip.data_ptr() if skip is not None else None, int(max_query), int(top_k),
                                                           float(max_dist), int(metric), int(pair_range[0]), int(pair_range[1]), S.data_ptr(),
                                                           self._stream()), "mme_page_similarity_pairs")
            return S
        self._check(self.lib.mme_page_similarity(self.h, emb.data_ptr(), N, d, area_pct.data_ptr(), valid.data_ptr(), page_offs.ctypes.data, P,
                                                 skip.data_ptr() if skip is not None else None, int(max_query), int(top_k), float(max_dist),
                                                 int(metric), int(bool(normalise)), S.data_ptr(), self._stream()), "mme_page_similarity")
        return S

    def cluster_pages(self, S, n_clusters=None, mode="reference_fallback"):
        """S: numpy f64 [P,P] (unit diagonal) or CUDA f64 tensor -> (labels int list, k, [(k, silhouette)...])."""
        t = self.torch
        dev = t.device(f"cuda:{self.device}")
        Sd = S if isinstance(S, t.Tensor) else t.from_numpy(np.ascontiguousarray(S, dtype=np.float64)).to(dev)
        Sd = Sd.contiguous()
        P = Sd.shape[0]
        if P < 2:
            raise MmeError("cluster_pages needs at least 2 pages")
        labels = t.empty(P, dtype=t.int32, device=dev)
        k = t.empty(1, dtype=t.int32, device=dev)
        scores = t.empty(16, dtype=t.float64, device=dev)
        m = {"reference_fallback": 0, "precomputed": 1}[mode]
        self._check(self.lib.mme_cluster_pages(self.h, Sd.data_ptr(), P, int(n_clusters or 0), m, labels.data_ptr(), k.data_ptr(),
                                               scores.data_ptr(), self._stream()), "mme_cluster_pages")
        sc = scores.cpu().numpy()
        return labels.cpu().numpy().tolist(), int(k.item()), [(i, float(sc[i])) for i in range(2, 16) if sc[i] == sc[i]]

    def set_neighbour_mode(self, mode):
        """0 / "auto", 1 / "block", 2 / "fused" (see mme.h)."""
        m = {"auto": 0, "block": 1, "fused": 2}.get(mode, mode)
        self._check(self.lib.mme_set_neighbour_mode(self.h, int(m)), "mme_set_neighbour_mode")

    def neighbours(self, emb_bf16, group=None, *, row0=0, nrows=None, fetch=30, top_n=10, keep_self=False,
                   min_sim=-float("inf"), max_sim=float("inf")):
        """Ranked neighbour lists of rows [row0, row0+nrows) of the normalised bf16 rows emb_bf16[N, d].

        group: optional int32[N] (CUDA tensor or array); rows with the query's group id are dropped.
        Returns (idx int32[nrows, top_n] padded with -1, sim float32[nrows, top_n]) CUDA tensors."""
        t = self.torch
        dev = t.device(f"cuda:{self.device}")
        e = emb_bf16.contiguous()
        if e.dtype != t.bfloat16 or e.dim() != 2 or e.device != dev:
            raise MmeError("neighbours: emb must be a 2-D bfloat16 tensor on this engine's device")
        N, d = e.shape
        nrows = N - row0 if nrows is None else int(nrows)
        g = None
        if group is not None:
            g = group if isinstance(group, t.Tensor) else t.from_numpy(np.ascontiguousarray(group, dtype=np.int32))
            g = g.to(device=dev, dtype=t.int32).contiguous()
            if g.numel() != N:
                raise MmeError("neighbours: group must have one id per row")
        idx = t.empty((max(nrows, 0), top_n), dtype=t.int32, device=dev)
        sim = t.empty((max(nrows, 0), top_n), dtype=t.float32, device=dev)
        self._check(self.lib.mme_neighbours(self.h, e.data_ptr(), N, d, g.data_ptr() if g is not None else None, int(row0), nrows,
                                            int(fetch), int(top_n), int(bool(keep_self)), float(min_sim), float(max_sim),
                                            idx.data_ptr(), sim.data_ptr(), self._stream()), "mme_neighbours")
        return idx, sim

    def gemm_bench(self, M, N, K, epilogue=0, variant=0, iters=10):
        """(avg ms, TFLOP/s) for one GEMM shape on random data."""
        ms = C.c_double(0)
        self._check(self.lib.mme_gemm_bench(self.h, M, N, K, epilogue, variant, iters, C.byref(ms)), "mme_gemm_bench")
        return ms.value, 2.0 * M * N * K / (ms.value * 1e-3) / 1e12

    def gemm_stamps(self, M, N, K):
        """uint64[256, 2, 16] in-kernel cycle stamps of the stamped GEMM build (see mme.h)."""
        st = np.zeros((256, 2, 16), dtype=np.uint64)
        self._check(self.lib.mme_gemm_stamps(self.h, M, N, K, st.ctypes.data), "mme_gemm_stamps")
        return st

    # ---- the one collective, without torch.distributed (mme.h: mme_comm_*, mme_allgather) ---------------
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        self._check(self.lib.mme_comm_unique_id(self.h, buf), "mme_comm_unique_id")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        comm = C.c_void_p()
        self._check(self.lib.mme_comm_init(self.h, buf, int(rank), int(world), C.byref(comm)), "mme_comm_init")
        return comm

    def comm_destroy(self, comm):
        self._check(self.lib.mme_comm_destroy(self.h, comm), "mme_comm_destroy")

    def allgather(self, comm, shard, world: int, out=None):
        """shard: bf16 CUDA tensor [rows, d] (same rows on every rank) -> [world * rows, d] on every rank."""
        t = self.torch
        assert shard.dtype == t.bfloat16 and shard.is_contiguous()
        rows, d = shard.shape
        if out is None:
            out = t.empty((world * rows, d), dtype=t.bfloat16, device=shard.device)
        self._check(self.lib.mme_allgather(self.h, comm, shard.data_ptr(), rows, d, out.data_ptr(), self._stream()), "mme_allgather")
        return out

    def attention_stamps(self, B, iters=5):
        """(avg ms of the product kernel, uint64[B, 8, 8] cycle stamps of the stamped build; see mme.h)."""
        st = np.zeros((B, 8, 8), dtype=np.uint64)
        ms = C.c_double(0)
        self._check(self.lib.mme_attention_stamps(self.h, int(B), int(iters), C.byref(ms), st.ctypes.data), "mme_attention_stamps")
        return ms.value, st

    # ---- timing ------------------------------------------------------------------------------------
    def profile(self, on: bool):
        self._check(self.lib.mme_profile_enable(self.h, int(on)), "mme_profile_enable")
        self._check(self.lib.mme_profile_reset(self.h), "mme_profile_reset")

    def profile_read(self):
        ms = (C.c_double * NUM_KERNEL_CLASSES)()
        cnt = (C.c_int64 * NUM_KERNEL_CLASSES)()
        rc = self.lib.mme_profile_read_sync(self.h, NUM_KERNEL_CLASSES, ms, cnt)
        self._check(min(rc, 0), "mme_profile_read_sync")
        return {KERNEL_CLASSES[i]: (ms[i], cnt[i]) for i in range(NUM_KERNEL_CLASSES)}
