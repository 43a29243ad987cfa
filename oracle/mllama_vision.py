"""ORACLE (test infrastructure only) -- fp32 CPU restatement of the Mllama vision tower.

The reference embeds with `MllamaForConditionalGeneration` (deprecated_package/embedder.py:75-79,117-126); its
image side is third-party code:

  transformers (requirements.txt:3; container pin 5.15.0) models/mllama/modeling_mllama.py, `MllamaVisionModel`
    patch embedding      Conv2d(3, 1280, k=14, s=14, bias=False) per 560 x 560 tile -> 1600 tokens
    pre-tile embedding   + tanh(gate) * E_pre[aspect_ratio_id][tile]          (patch tokens only)
    class token          prepended per tile -> 1601 tokens
    position embedding   + (1 - tanh(g)) * E_pos[token] + tanh(g) * E_tile[aspect_ratio_id][tile][token]
    layernorm_pre; tokens padded 1601 -> 1608 with zeros; the 4 tiles form ONE sequence of 6432 tokens
    attention mask       additive finfo.min where BOTH the query and the key are padding (a padding tile or one of
                         the 7 padding tokens of a tile); every other pair attends -- including valid -> padding
    32 local layers      x += o_proj(attn(LN(x)));  x += fc2(gelu_erf(fc1(LN(x))))     (q/k/v/o without bias, eps 1e-5)
    layernorm_post, + tanh(gate) * E_post[aspect_ratio_id][tile]
    8 global layers      the same block with the two branches scaled by tanh(gate_attn) / tanh(gate_ffn)
    output               cat(final state, stack(states after layers 3, 7, 15, 23, 30, dim=-1).flatten) -> 7680 per token,
                         padding tokens dropped: [B, 1, 4, 1601, 7680]

Pinning: `vision_forward` is checked against transformers' `MllamaVisionModel(MllamaVisionConfig(image_size=560))`
holding the same seeded weights by tests/golden/make_golden.py (`golden_tile_vit`, container only); the recorded rows
are committed as tests/golden/tile_vit_cases.npz and re-checked by tests/test_oracle_pins.py.  No pretrained vectors
ship with the reference, so absolute values w.r.t. the real mmE5 checkpoint stay "parity unpinned".
"""
from __future__ import annotations

import math

import numpy as np
import torch

from multimodal_embeddings_amd.weights import TILE_VIT, TileViTGeometry


def _t(w, name):
    return torch.from_numpy(np.ascontiguousarray(w[name], dtype=np.float32))


def _ln(x, g, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), g, b, eps)


def padding_flags(num_tiles: int, geom: TileViTGeometry = TILE_VIT) -> torch.Tensor:
    """bool [max_tiles * padded_patches]: True for the tokens the attention mask treats as padding."""
    tok = torch.arange(geom.padded_patches)
    tile = torch.arange(geom.max_num_tiles)
    pad = (tok[None, :] >= geom.num_patches) | (tile[:, None] >= num_tiles)
    return pad.reshape(-1)


def _block(x, w, p, pad, geom, gate_attn=None, gate_ffn=None):
    D, H, dh = geom.hidden_size, geom.num_heads, geom.head_dim
    h = _ln(x, _t(w, p + "input_layernorm.weight"), _t(w, p + "input_layernorm.bias"), geom.norm_eps)
    q = (h @ _t(w, p + "self_attn.q_proj.weight").T).reshape(-1, H, dh).transpose(0, 1)
    k = (h @ _t(w, p + "self_attn.k_proj.weight").T).reshape(-1, H, dh).transpose(0, 1)
    v = (h @ _t(w, p + "self_attn.v_proj.weight").T).reshape(-1, H, dh).transpose(0, 1)
    out = torch.empty_like(q)
    both = pad[:, None] & pad[None, :]
    for hd in range(H):  # one head at a time: the [T, T] score matrix of 6432 tokens is 165 MB
        s = (q[hd] @ k[hd].T) * dh ** -0.5
        s = s.masked_fill(both, torch.finfo(torch.float32).min)
        out[hd] = torch.softmax(s, dim=-1) @ v[hd]
    a = out.transpose(0, 1).reshape(-1, D) @ _t(w, p + "self_attn.o_proj.weight").T
    x = x + (a if gate_attn is None else math.tanh(gate_attn) * a)
    h = _ln(x, _t(w, p + "post_attention_layernorm.weight"), _t(w, p + "post_attention_layernorm.bias"), geom.norm_eps)
    h = torch.nn.functional.gelu(h @ _t(w, p + "mlp.fc1.weight").T + _t(w, p + "mlp.fc1.bias"))
    h = h @ _t(w, p + "mlp.fc2.weight").T + _t(w, p + "mlp.fc2.bias")
    return x + (h if gate_ffn is None else math.tanh(gate_ffn) * h)


@torch.no_grad()
def vision_forward(pixel_values, aspect_ratio_id: int, num_tiles: int, w: dict, geom: TileViTGeometry = TILE_VIT) -> np.ndarray:
    """pixel_values f32 [max_tiles, 3, S, S] of ONE image (what `mme_preprocess_tiles` / the Mllama processor produce),
    its aspect-ratio id and tile count -> last_hidden_state f32 [max_tiles, 1601, 7680]."""
    D, T, NP, PP, P = geom.hidden_size, geom.max_num_tiles, geom.num_patches, geom.padded_patches, geom.patch_size
    px = torch.from_numpy(np.ascontiguousarray(pixel_values, dtype=np.float32))
    g = geom.image_size // P
    patches = px.reshape(T, geom.num_channels, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(T, g * g, geom.patch_dim)
    x = patches @ _t(w, "patch_embedding.weight").reshape(D, -1).T  # [T, 1600, D]
    pre = _t(w, "pre_tile_positional_embedding.embedding.weight")[aspect_ratio_id].reshape(T, 1, D)
    x = x + pre * math.tanh(float(w["pre_tile_positional_embedding.gate"][0]))
    x = torch.cat([_t(w, "class_embedding").expand(T, 1, D), x], dim=1)  # [T, 1601, D]
    gp = math.tanh(float(w["gated_positional_embedding.gate"][0]))
    x = x + (1.0 - gp) * _t(w, "gated_positional_embedding.embedding")[None]
    x = x + gp * _t(w, "gated_positional_embedding.tile_embedding.weight")[aspect_ratio_id].reshape(T, NP, D)
    x = _ln(x, _t(w, "layernorm_pre.weight"), _t(w, "layernorm_pre.bias"), 1e-5)
    x = torch.nn.functional.pad(x, (0, 0, 0, PP - NP)).reshape(T * PP, D)
    pad = padding_flags(num_tiles, geom)
    keep = []
    before = getattr(geom, "intermediate_save_point", "after") == "before"
    for i in range(geom.num_layers):
        if before and i in geom.intermediate_layers:  # the state ENTERING layer i (include/mme.h, MME_TILE_SAVE_BEFORE_LAYER)
            keep.append(x)
        x = _block(x, w, f"transformer.layers.{i}.", pad, geom)
        if not before and i in geom.intermediate_layers:  # transformers 5.15: encoder_states appended after each layer
            keep.append(x)
    x = _ln(x, _t(w, "layernorm_post.weight"), _t(w, "layernorm_post.bias"), 1e-5).reshape(T, PP, D)
    post = _t(w, "post_tile_positional_embedding.embedding.weight")[aspect_ratio_id].reshape(T, 1, D)
    x = (x + post * math.tanh(float(w["post_tile_positional_embedding.gate"][0]))).reshape(T * PP, D)
    for i in range(geom.num_global_layers):
        p = f"global_transformer.layers.{i}."
        x = _block(x, w, p, pad, geom, float(w[p + "gate_attn"][0]), float(w[p + "gate_ffn"][0]))
    inter = torch.stack(keep, dim=-1).reshape(T, PP, D * len(keep))  # feature index d * 5 + layer
    out = torch.cat([x.reshape(T, PP, D), inter], dim=-1)[:, :NP]
    return out.numpy()


def pooled_embedding(last_hidden_state: np.ndarray) -> np.ndarray:
    """The D-dim vector of a crop for the compare stage: the class token of tile 0, L2-normalised
    (the pooling rule of embedder.py:17-34 -- one token row, F.normalize -- applied to the tower's output)."""
    v = last_hidden_state[0, 0].astype(np.float64)
    return (v / max(np.linalg.norm(v), 1e-12)).astype(np.float32)
