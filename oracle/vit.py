"""ORACLE (test infrastructure only) -- fp32 CPU ViT-B/16 forward + pooling.

Restates, with plain torch CPU fp32 tensor ops, the dense forward the reference
runs per crop at deprecated_package/embedder.py:124-129
(`model(**inputs, output_hidden_states=True)` then `last_pooling`) for the
re-scoped encoder BASELINE.json names (ViT-style patchify + transformer, ViT-B/16
@224).  The encoder arithmetic itself is third-party:

  transformers (requirements.txt:3 `>=4.34.0`; container pin 5.15.0)
  models/vit/modeling_vit.py
    :42-70    patch embedding = Conv2d(k=16, s=16) -> flatten -> transpose
    :72-160   [CLS] prepend + learned position embeddings
    :164-189  eager attention: softmax(QK^T * dh^-0.5) in f32, then PV
    :192-239  q/k/v/o projections (bias=True)
    :241-255  MLP fc1 -> erf-GELU -> fc2
    :257-287  pre-LN residual block
    :336-400  final LayerNorm (eps 1e-12)

and the pooling is the reference's own `last_pooling`
(deprecated_package/embedder.py:17-34): gather one token row per sequence
(`attention_mask.sum(1) - 1`), L2-normalise with torch.nn.functional.normalize.

Pinning: `vit_forward` is checked against `transformers.ViTModel` loaded with the
same seeded weights, and `last_pooling` against the reference function imported
from /root/reference, by tests/golden/make_golden.py (container only); outputs
are committed under tests/golden/ and re-checked by tests/test_oracle_pins.py.
No pretrained vectors ship with the reference, so absolute
embedding values are "parity unpinned" w.r.t. the real mmE5 checkpoint.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from multimodal_embeddings_amd.weights import VIT_B16, ViTGeometry


def _t(w, name):
    return torch.from_numpy(np.ascontiguousarray(w[name], dtype=np.float32))


def last_pooling(last_hidden_state: torch.Tensor, attention_mask: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """embedder.py:17-34 restated."""
    sequence_lengths = attention_mask.sum(dim=1) - 1
    rows = torch.arange(last_hidden_state.shape[0])
    reps = last_hidden_state[rows, sequence_lengths]
    if normalize:
        reps = torch.nn.functional.normalize(reps, p=2, dim=-1)
    return reps


def layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


@torch.no_grad()
def vit_hidden_states(patches: torch.Tensor, w: dict, geom: ViTGeometry = VIT_B16, final_ln: bool = True) -> torch.Tensor:
    """f32 patches [B, 196, 768] (c,ky,kx order) -> last hidden state [B, 197, 768]."""
    B = patches.shape[0]
    D, H, dh = geom.hidden_size, geom.num_heads, geom.head_dim
    wp = _t(w, "embeddings.patch_embeddings.projection.weight").reshape(D, -1)
    x = patches @ wp.T + _t(w, "embeddings.patch_embeddings.projection.bias")
    cls = _t(w, "embeddings.cls_token").reshape(1, 1, D).expand(B, 1, D)
    x = torch.cat([cls, x], dim=1) + _t(w, "embeddings.position_embeddings").reshape(1, geom.seq_len, D)
    for i in range(geom.num_layers):
        p = f"layers.{i}."
        h = layer_norm(x, _t(w, p + "layernorm_before.weight"), _t(w, p + "layernorm_before.bias"), geom.layer_norm_eps)
        q = h @ _t(w, p + "attention.q_proj.weight").T + _t(w, p + "attention.q_proj.bias")
        k = h @ _t(w, p + "attention.k_proj.weight").T + _t(w, p + "attention.k_proj.bias")
        v = h @ _t(w, p + "attention.v_proj.weight").T + _t(w, p + "attention.v_proj.bias")
        q = q.view(B, -1, H, dh).transpose(1, 2)
        k = k.view(B, -1, H, dh).transpose(1, 2)
        v = v.view(B, -1, H, dh).transpose(1, 2)
        s = torch.softmax((q @ k.transpose(2, 3)) * (dh ** -0.5), dim=-1)
        a = (s @ v).transpose(1, 2).reshape(B, -1, D)
        x = x + (a @ _t(w, p + "attention.o_proj.weight").T + _t(w, p + "attention.o_proj.bias"))
        h = layer_norm(x, _t(w, p + "layernorm_after.weight"), _t(w, p + "layernorm_after.bias"), geom.layer_norm_eps)
        h = gelu_erf(h @ _t(w, p + "mlp.fc1.weight").T + _t(w, p + "mlp.fc1.bias"))
        x = x + (h @ _t(w, p + "mlp.fc2.weight").T + _t(w, p + "mlp.fc2.bias"))
    if final_ln:
        x = layer_norm(x, _t(w, "layernorm.weight"), _t(w, "layernorm.bias"), geom.layer_norm_eps)
    return x


@torch.no_grad()
def vit_embed(patches, w: dict, geom: ViTGeometry = VIT_B16, pool: str = "cls", batch: int = 16) -> np.ndarray:
    """patches f32 [B,196,768] -> L2-normalised f32 [B,768] embeddings.

    pool="cls": token 0 (the ViT summary token).  pool="last": token T-1, i.e.
    `last_pooling` with an all-ones attention mask, as the reference pools a
    decoder sequence (embedder.py:29-31).
    """
    patches = torch.as_tensor(np.asarray(patches, dtype=np.float32))
    outs = []
    for s in range(0, patches.shape[0], batch):
        hs = vit_hidden_states(patches[s : s + batch], w, geom)
        n = hs.shape[0]
        if pool == "cls":
            mask = torch.zeros(n, geom.seq_len, dtype=torch.long)
            mask[:, 0] = 1
        elif pool == "last":
            mask = torch.ones(n, geom.seq_len, dtype=torch.long)
        else:
            raise ValueError(pool)
        outs.append(last_pooling(hs, mask))
    return torch.cat(outs).numpy()
