"""Deterministic synthetic ViT-B/16 weights (SURVEY.md §8d "Synthetic inputs").

The reference loads pretrained weights by NAME over the network
(deprecated_package/embedder.py:75-79), which is unavailable offline, so the
encoder of this build runs on seeded synthetic weights.  The generator is a
counter-based integer hash, so the very same tensors are produced in the build
container, on the GPU box and inside the golden-vector script without shipping
343 MB of floats:

    value(seed, tensor_id, i) = bf16_round( offset + std * z ),
    z = (sum of twelve 16-bit lanes of splitmix64 words - 393210) / 65536

(an Irwin-Hall approximation of N(0,1); every step is integer or a single
correctly-rounded IEEE operation, so no libm call can differ between hosts).
All values are bf16-representable: the reference itself runs its encoder with
``torch_dtype=torch.bfloat16`` (embedder.py:78), and this lets the fp32 oracle
and the bf16 MFMA path consume bit-identical parameters.

Tensor names follow the Hugging Face ViT checkpoint layout
(transformers/models/vit/modeling_vit.py; `ViTModel(add_pooling_layer=False)`)
so the same dict loads into that class with ``load_state_dict`` when the golden
vectors are generated.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


@dataclass(frozen=True)
class ViTGeometry:
    """ViT-B/16 @224 (transformers ViTConfig defaults)."""

    image_size: int = 224
    patch_size: int = 16
    num_channels: int = 3
    hidden_size: int = 768
    num_layers: int = 12
    num_heads: int = 12
    intermediate_size: int = 3072
    layer_norm_eps: float = 1e-12

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def seq_len(self) -> int:
        return self.num_patches + 1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.patch_size * self.patch_size


VIT_B16 = ViTGeometry()


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK64
        z = z ^ (z >> np.uint64(31))
    return z


def counter_u64(seed: int, stream: int, n: int, word: int = 0) -> np.ndarray:
    """n hashed 64-bit words for counters (seed, stream, i, word)."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = (
            np.uint64(seed & 0xFFFF) * np.uint64(1 << 48)
            + np.uint64(stream & 0xFFFF) * np.uint64(1 << 32)
        )
        key = _splitmix64(np.uint64(base) + np.uint64(word) * np.uint64(0xD1B54A32D192ED03))
        return _splitmix64(key ^ (i * np.uint64(0x2545F4914F6CDD1D)))


def irwin_hall_normal(seed: int, stream: int, n: int) -> np.ndarray:
    """Approximate N(0,1) float32 samples with exact integer provenance."""
    total = np.zeros(n, dtype=np.int64)
    for word in range(3):
        w = counter_u64(seed, stream, n, word)
        for lane in range(4):
            total += ((w >> np.uint64(16 * lane)) & np.uint64(0xFFFF)).astype(np.int64)
    # mean of twelve U{0..65535} is 12*32767.5 = 393210; var = 12*(65536^2-1)/12
    return ((total - 393210).astype(np.float32)) * np.float32(1.0 / 65536.0)


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even f32 -> bf16, returned as f32 (finite inputs only)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    with np.errstate(over="ignore"):
        r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """bf16 bit pattern (uint16) of finite f32 values, round-to-nearest-even."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    with np.errstate(over="ignore"):
        r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def vit_tensor_specs(geom: ViTGeometry = VIT_B16):
    """(name, shape, kind) in a fixed order; kind in {matrix, bias, gamma}."""
    D, F, P = geom.hidden_size, geom.intermediate_size, geom.patch_size
    specs = [
        ("embeddings.cls_token", (1, 1, D), "matrix"),
        ("embeddings.position_embeddings", (1, geom.seq_len, D), "matrix"),
        ("embeddings.patch_embeddings.projection.weight", (D, geom.num_channels, P, P), "matrix"),
        ("embeddings.patch_embeddings.projection.bias", (D,), "bias"),
    ]
    for i in range(geom.num_layers):
        p = f"layers.{i}."
        specs += [
            (p + "layernorm_before.weight", (D,), "gamma"),
            (p + "layernorm_before.bias", (D,), "bias"),
            (p + "attention.q_proj.weight", (D, D), "matrix"),
            (p + "attention.q_proj.bias", (D,), "bias"),
            (p + "attention.k_proj.weight", (D, D), "matrix"),
            (p + "attention.k_proj.bias", (D,), "bias"),
            (p + "attention.v_proj.weight", (D, D), "matrix"),
            (p + "attention.v_proj.bias", (D,), "bias"),
            (p + "attention.o_proj.weight", (D, D), "matrix"),
            (p + "attention.o_proj.bias", (D,), "bias"),
            (p + "layernorm_after.weight", (D,), "gamma"),
            (p + "layernorm_after.bias", (D,), "bias"),
            (p + "mlp.fc1.weight", (F, D), "matrix"),
            (p + "mlp.fc1.bias", (F,), "bias"),
            (p + "mlp.fc2.weight", (D, F), "matrix"),
            (p + "mlp.fc2.bias", (D,), "bias"),
        ]
    specs += [("layernorm.weight", (D,), "gamma"), ("layernorm.bias", (D,), "bias")]
    return specs


def make_vit_weights(
    seed: int = 1,
    geom: ViTGeometry = VIT_B16,
    std: float = 0.02,
    trained_like: bool = True,
) -> dict[str, np.ndarray]:
    """Seeded synthetic weights, f32 arrays holding bf16-representable values.

    ``trained_like=False`` reproduces the Hugging Face initialiser exactly as
    SURVEY.md §8d words it (biases 0, LayerNorm gamma 1 / beta 0).  The default
    perturbs biases and LayerNorm parameters as well, so that every bias-add and
    affine path of the kernels carries non-trivial data in the parity tests; the
    arithmetic cost is identical.
    """
    out: dict[str, np.ndarray] = {}
    for tid, (name, shape, kind) in enumerate(vit_tensor_specs(geom)):
        n = int(np.prod(shape))
        if kind == "matrix" or trained_like:
            z = irwin_hall_normal(seed, tid, n) * np.float32(std)
            if kind == "gamma":
                z = z + np.float32(1.0)
        else:
            z = np.full(n, 1.0 if kind == "gamma" else 0.0, dtype=np.float32)
        out[name] = round_to_bf16(z).reshape(shape)
    return out


def synthetic_crops(n: int, seed: int = 0, size: int = 224, start: int = 0) -> np.ndarray:
    """uint8[n, size, size, 3] i.i.d. uniform 0..255 (SURVEY.md §8d, C2/C4 inputs).

    ``start`` offsets the crop index so that a rank can generate only its shard.
    """
    per = size * size * 3
    assert per % 8 == 0
    out = np.empty((n, per), dtype=np.uint8)
    words = per // 8
    for k in range(n):
        w = counter_u64(seed, 0x7000 + ((start + k) >> 16), words, (start + k) & 0xFFFF)
        out[k] = w.view(np.uint8)
    return out.reshape(n, size, size, 3)


def synthetic_page_structure(pages: int = 512, per_page: int = 128, seed: int = 2, duplicated_prefixes: int = 0):
    """The page structure of config C5: `pages` pages of `per_page` regions, page id = idx // per_page.

    Returns (area_percentage f64 [N] on the reference's 0..100 scale (region_processor.py:89-93), page_offs int32
    [pages + 1], names).  Area fractions are seeded log-uniform in [1e-4, 0.2], scaled down where a page would sum to
    more than 1; integer provenance (counter hash), so every box and every rank builds the same table.  Page names have
    distinct first-20-character prefixes (no same-prefix skips, wrc:179-186) except that the first
    `duplicated_prefixes` odd pages repeat the prefix of the page before them."""
    n = pages * per_page
    u = (counter_u64(seed, 0x6100, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    frac = np.exp(np.log(1e-4) + u * (np.log(0.2) - np.log(1e-4))).reshape(pages, per_page)
    tot = frac.sum(axis=1, keepdims=True)
    frac = np.where(tot > 1.0, frac / tot, frac)
    names = [f"{p:04d} synthetic page of the C5 set.png" for p in range(pages)]
    for k in range(duplicated_prefixes):
        names[2 * k + 1] = names[2 * k][:20] + f" second scan {k}.png"
    return (frac.reshape(-1) * 100.0), (np.arange(pages + 1, dtype=np.int64) * per_page).astype(np.int32), names


# ---- the reference's own vision-tower geometry --------------------------------------------------
@dataclass(frozen=True)
class TileViTGeometry:
    """Mllama vision tower as the reference's checkpoint configures it (`config.py:58`, transformers
    `configuration_mllama.py:61-82` at image_size 560): <= 4 tiles of 560 x 560, patch 14, 1 + 1600 tokens per tile
    (padded to 1608), 1280-d, 16 heads of 80, MLP 5120, 32 local + 8 gated global layers, the outputs of five
    intermediate layers concatenated to the final one -> 7680-d."""

    image_size: int = 560
    patch_size: int = 14
    num_channels: int = 3
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 5120
    num_layers: int = 32
    num_global_layers: int = 8
    max_num_tiles: int = 4
    max_aspect_ratio_id: int = 8
    intermediate_layers: tuple = (3, 7, 15, 23, 30)
    # "after": index i names the OUTPUT of local layer i (transformers 5.15, what the fixtures pin);
    # "before": the state entering layer i (encoders that record before running a layer; include/mme.h)
    intermediate_save_point: str = "after"
    norm_eps: float = 1e-5

    @property
    def num_patches(self) -> int:  # tokens of a tile, class token included
        return (self.image_size // self.patch_size) ** 2 + 1

    @property
    def padded_patches(self) -> int:
        return (self.num_patches + 7) // 8 * 8

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.patch_size * self.patch_size

    @property
    def output_dim(self) -> int:
        return self.hidden_size * (1 + len(self.intermediate_layers))


TILE_VIT = TileViTGeometry()


def tile_vit_tensor_specs(geom: TileViTGeometry = TILE_VIT):
    """(name, shape, kind, scale) in a fixed order, Hugging Face `MllamaVisionModel` state-dict names."""
    D, F, P, T = geom.hidden_size, geom.intermediate_size, geom.patch_size, geom.max_num_tiles
    A = geom.max_aspect_ratio_id + 1
    emb = D ** -0.5
    specs = [
        ("class_embedding", (D,), "matrix", emb),
        ("patch_embedding.weight", (D, geom.num_channels, P, P), "matrix", 0.02),
        ("gated_positional_embedding.gate", (1,), "gate", 0.6),
        ("gated_positional_embedding.embedding", (geom.num_patches, D), "matrix", emb),
        ("gated_positional_embedding.tile_embedding.weight", (A, T * geom.num_patches * D), "matrix", 0.02),
        ("pre_tile_positional_embedding.gate", (1,), "gate", 0.4),
        ("pre_tile_positional_embedding.embedding.weight", (A, T * D), "matrix", 0.02),
        ("post_tile_positional_embedding.gate", (1,), "gate", -0.5),
        ("post_tile_positional_embedding.embedding.weight", (A, T * D), "matrix", 0.02),
        ("layernorm_pre.weight", (D,), "gamma", 0.02),
        ("layernorm_pre.bias", (D,), "bias", 0.02),
        ("layernorm_post.weight", (D,), "gamma", 0.02),
        ("layernorm_post.bias", (D,), "bias", 0.02),
    ]
    for stack, count, gated in (("transformer", geom.num_layers, False), ("global_transformer", geom.num_global_layers, True)):
        for i in range(count):
            p = f"{stack}.layers.{i}."
            if gated:
                specs += [(p + "gate_attn", (1,), "gate", 0.7853981633974483), (p + "gate_ffn", (1,), "gate", 0.7853981633974483)]
            specs += [
                (p + "self_attn.q_proj.weight", (D, D), "matrix", 0.02),
                (p + "self_attn.k_proj.weight", (D, D), "matrix", 0.02),
                (p + "self_attn.v_proj.weight", (D, D), "matrix", 0.02),
                (p + "self_attn.o_proj.weight", (D, D), "matrix", 0.02),
                (p + "mlp.fc1.weight", (F, D), "matrix", 0.02),
                (p + "mlp.fc1.bias", (F,), "bias", 0.02),
                (p + "mlp.fc2.weight", (D, F), "matrix", 0.02),
                (p + "mlp.fc2.bias", (D,), "bias", 0.02),
                (p + "input_layernorm.weight", (D,), "gamma", 0.02),
                (p + "input_layernorm.bias", (D,), "bias", 0.02),
                (p + "post_attention_layernorm.weight", (D,), "gamma", 0.02),
                (p + "post_attention_layernorm.bias", (D,), "bias", 0.02),
            ]
    return specs


def hashed_normal4(seed: int, stream: int, n: int, start: int = 0) -> np.ndarray:
    """Approximate N(0,1) f32 samples for element counters start .. start+n: the four 16-bit lanes of ONE splitmix64 word
    summed (Irwin-Hall, 4 terms), centred and scaled to unit variance.  Integer provenance like `irwin_hall_normal`,
    a third of its cost -- the vision tower has 860 M parameters to fill."""
    i = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64(seed & 0xFFFF) * np.uint64(1 << 48) + np.uint64(stream & 0xFFFFFFFF) * np.uint64(1 << 16)
        w = _splitmix64(_splitmix64(np.uint64(base)) ^ (i * np.uint64(0x2545F4914F6CDD1D)))
    total = (w & np.uint64(0xFFFF)).astype(np.int32)
    for lane in (1, 2, 3):
        total += ((w >> np.uint64(16 * lane)) & np.uint64(0xFFFF)).astype(np.int32)
    # four U{0..65535}: mean 131070, variance 4 * (65536^2 - 1) / 12 -> sigma = 37837.2247...
    return (total - 131070).astype(np.float32) * np.float32(1.0 / 37837.224732)


def make_tile_vit_weights(seed: int = 2, geom: TileViTGeometry = TILE_VIT, threads: int = 8) -> dict[str, np.ndarray]:
    """Seeded synthetic weights of the Mllama vision tower, f32 arrays holding bf16-representable values (the reference
    runs the tower in bf16, embedder.py:78).  A counter-based generator: identical tensors in the build container (where
    they are loaded into transformers' `MllamaVisionModel` to make the golden vectors) and on the GPU box.  Gates (the
    tanh-gated embeddings and the gated global layers) get fixed non-zero values so that every gated path carries data;
    matrices N(0, scale), LayerNorm gamma 1 + N(0, 0.02), biases N(0, 0.02)."""
    from concurrent.futures import ThreadPoolExecutor

    out: dict[str, np.ndarray] = {}
    jobs = []
    step = 1 << 22  # elements per block: bounds the uint64 temporaries and gives the thread pool even work
    for tid, (name, shape, kind, scale) in enumerate(tile_vit_tensor_specs(geom)):
        n = int(np.prod(shape))
        if kind == "gate":
            out[name] = round_to_bf16(np.full(n, scale, dtype=np.float32)).reshape(shape)
            continue
        buf = np.empty(n, dtype=np.float32)
        out[name] = buf.reshape(shape)
        jobs += [(buf, tid, o, min(step, n - o), scale, kind) for o in range(0, n, step)]

    def fill(job):
        buf, tid, o, m, scale, kind = job
        z = hashed_normal4(seed, tid, m, start=o) * np.float32(scale)
        if kind == "gamma":
            z += np.float32(1.0)
        buf[o : o + m] = round_to_bf16(z)

    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(fill, jobs))
    return out
