"""ORACLE (test infrastructure only) -- crop -> resize -> pad -> normalise -> patchify.

CPU restatement of what the reference does to every region crop before the
encoder sees it.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product path never does.

Reference path restated here
  * deprecated_package/doclayout_detector.py:178-190  int()-truncated bbox crop
  * deprecated_package/embedder.py:104-121           open -> processor(images=[image])
  * third-party transformers (requirements.txt:3; container pin 5.15.0)
    models/mllama/image_processing_pil_mllama.py
      :246-295 get_image_size_fit_to_canvas   (aspect-preserving fit)
      :392-429 pad                            (zero pad right/bottom, BEFORE normalise)
      :483-541 _preprocess                    (resize -> pad -> *1/255 -> (x-mean)/std)
    image_transforms.py:89-125 rescale  (u8 -> f64 * scale -> f32)
    image_transforms.py:384-440 normalize ((x - mean_f32) / std_f32 in f32)
  * third-party Pillow 12.2.0 `Image.resize(..., BILINEAR)` = libImaging/Resample.c
    (8 bits per channel path): separable, horizontal pass then vertical pass,
    triangle filter whose support grows with the down-scale factor, coefficients
    quantised to 22 fractional bits, uint8 rounding after each pass.

Pinning: `pil_bilinear_resize_u8` is checked bit-for-bit against Pillow itself and
`preprocess_crop` against MllamaImageProcessorPil(size=224, max_image_tiles=1) in
tests/test_oracle_pins.py (container and GPU box both carry Pillow/transformers),
and against committed fixtures in tests/golden/ made by tests/golden/make_golden.py.

The re-scoped encoder is ViT-B/16 @224 (BASELINE.json north_star), so tile = 224
and max_image_tiles = 1; mean/std default to the CLIP constants the real mmE5
checkpoint ships (SURVEY.md §8c "Preprocessing oracle").
"""
from __future__ import annotations

import math

import numpy as np

TILE = 224
PATCH = 16
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2  # Resample.c


def crop_box_int(box) -> tuple[int, int, int, int]:
    """doclayout_detector.py:179 `x_min, y_min, x_max, y_max = map(int, box)`."""
    x0, y0, x1, y1 = (int(v) for v in box)
    return x0, y0, x1, y1


def fit_to_canvas(h: int, w: int, tile: int = TILE) -> tuple[int, int]:
    """image_processing_pil_mllama.py:246-295 with canvas == tile (one tile).

    np.clip(x, tile, tile) == tile, so target = tile on both axes.
    """
    scale_h = tile / h
    scale_w = tile / w
    if scale_w < scale_h:
        new_w = tile
        new_h = min(math.floor(h * scale_w) or 1, tile)
    else:
        new_h = tile
        new_w = min(math.floor(w * scale_h) or 1, tile)
    return new_h, new_w


def resample_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter.

    Returns (xmin[out], count[out], kk[out, ksize] int64 fixed-point weights).
    """
    scale = in_size / out_size  # (double)(in1 - in0) / outSize, box = whole image
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale  # bilinear support = 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int64)
    cnt = np.zeros(out_size, dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        if lo < 0:
            lo = 0
        hi = int(center + support + 0.5)
        if hi > in_size:
            hi = in_size
        n = hi - lo
        x = np.arange(n, dtype=np.float64)
        t = np.abs((x + lo - center + 0.5) * ss)
        w = np.where(t < 1.0, 1.0 - t, 0.0)
        # Resample.c accumulates ww sequentially; restate that order exactly
        acc = 0.0
        for v in w:
            acc += v
        ww = acc
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        xmin[xx] = lo
        cnt[xx] = n
        kk[xx, :n] = q
    return xmin, cnt, kk


def _coeff_matrix(in_size: int, out_size: int) -> np.ndarray:
    xmin, cnt, kk = resample_coeffs(in_size, out_size)
    m = np.zeros((out_size, in_size), dtype=np.float64)
    for xx in range(out_size):
        m[xx, xmin[xx] : xmin[xx] + cnt[xx]] = kk[xx, : cnt[xx]]
    return m


def _apply_axis0(m: np.ndarray, img: np.ndarray) -> np.ndarray:
    """out[o, ...] = clip8((2^21 + sum_i m[o,i]*img[i,...]) >> 22), exact in f64.

    |sum| < 2^22 * 255 * (1 + ksize * 2^-23) < 2^31, far inside f64's 2^53.
    """
    shp = img.shape
    acc = m @ img.reshape(shp[0], -1).astype(np.float64)
    acc = acc.astype(np.int64) + (1 << (PRECISION_BITS - 1))
    out = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out.reshape((m.shape[0],) + shp[1:])


def pil_bilinear_resize_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """Bit-exact restatement of PIL `Image.resize((new_w,new_h), BILINEAR)` on u8 HWC.

    Resample.c ImagingResample: horizontal pass first (only when the width
    changes), then vertical pass (only when the height changes); identical size
    returns a copy.
    """
    h, w = img.shape[:2]
    out = img
    if new_w != w:
        m = _coeff_matrix(w, new_w)
        out = np.swapaxes(_apply_axis0(m, np.swapaxes(out, 0, 1)), 0, 1)
    if new_h != h:
        m = _coeff_matrix(h, new_h)
        out = _apply_axis0(m, out)
    return np.ascontiguousarray(out)


def normalise_lut(mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """f32[3,256]: u8 -> ((f32)(f64(u) * (1/255)) - mean_f32) / std_f32.

    image_transforms.py:89-125 (rescale through f64, cast f32) then
    image_transforms.py:384-440 (f32 subtract, f32 divide).
    """
    u = np.arange(256, dtype=np.float64)
    x = (u * (1 / 255)).astype(np.float32)
    m = np.array(mean, dtype=np.float32)
    s = np.array(std, dtype=np.float32)
    return ((x[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


def preprocess_crop(img: np.ndarray, mean=CLIP_MEAN, std=CLIP_STD, tile: int = TILE) -> np.ndarray:
    """u8[h,w,3] crop -> f32[3,tile,tile] pixel_values (one tile).

    Order per image_processing_pil_mllama.py:505-514: resize, zero-pad to the
    canvas, rescale, normalise -- so padding ends up at (0-mean)/std, not 0.
    """
    h, w = img.shape[:2]
    new_h, new_w = fit_to_canvas(h, w, tile)
    small = pil_bilinear_resize_u8(img, new_h, new_w)
    canvas = np.zeros((tile, tile, 3), dtype=np.uint8)
    canvas[:new_h, :new_w] = small
    lut = normalise_lut(mean, std)
    out = np.empty((3, tile, tile), dtype=np.float32)
    for c in range(3):
        out[c] = lut[c][canvas[:, :, c]]
    return out


def patchify(pixel_values: np.ndarray, patch: int = PATCH) -> np.ndarray:
    """f32[3,H,W] -> f32[(H/p)*(W/p), 3*p*p] in conv-weight order (c, ky, kx).

    Row-major patch index = py*(W/p)+px, matching `Conv2d(stride=p)` followed by
    `.flatten(2).transpose(1, 2)` (transformers models/vit/modeling_vit.py:42-70).
    """
    c, h, w = pixel_values.shape
    gh, gw = h // patch, w // patch
    x = pixel_values.reshape(c, gh, patch, gw, patch)
    return np.ascontiguousarray(x.transpose(1, 3, 0, 2, 4)).reshape(gh * gw, c * patch * patch)


def preprocess_to_patches(img: np.ndarray, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    return patchify(preprocess_crop(img, mean, std))


# ---- Mllama multi-tile preprocessing ------------------------------------------
# Restates transformers models/mllama/image_processing_pil_mllama.py as the reference's
# `processor(images=[image])` runs it with the checkpoint's geometry (tile 560, up to 4 tiles).


def supported_aspect_ratios(max_tiles: int):
    """:216-243 -- every (a, b) with a*b <= max_tiles, a outer, b inner."""
    return [(a, b) for a in range(1, max_tiles + 1) for b in range(1, max_tiles + 1) if a * b <= max_tiles]


def optimal_tiled_canvas(h: int, w: int, max_tiles: int, tile: int) -> tuple[int, int]:
    """:299-355 -- canvas (height, width) in pixels: the smallest upscaling if any canvas fits the image
    whole, else the largest downscaling; ties -> smallest area, then first in list order."""
    arr = supported_aspect_ratios(max_tiles)
    best = None
    scales = [min((a * tile) / h, (b * tile) / w) if True else 0 for a, b in arr]
    # np.where(scale_w > scale_h, scale_h, scale_w) == min on finite values
    up = [s for s in scales if s >= 1]
    sel = min(up) if up else max(s for s in scales if s < 1)
    for (a, b), s in zip(arr, scales):
        if s == sel:
            area = (a * tile) * (b * tile)
            if best is None or area < best[0]:
                best = (area, a * tile, b * tile)
    return best[1], best[2]


def fit_to_canvas_general(h: int, w: int, canvas_h: int, canvas_w: int, tile: int) -> tuple[int, int]:
    """:246-295 -- aspect-preserving size inside the canvas, never below one tile on the binding axis."""
    target_w = min(max(w, tile), canvas_w)
    target_h = min(max(h, tile), canvas_h)
    scale_h = target_h / h
    scale_w = target_w / w
    if scale_w < scale_h:
        return min(math.floor(h * scale_w) or 1, target_h), target_w
    return target_h, min(math.floor(w * scale_h) or 1, target_w)


def preprocess_tiles(img: np.ndarray, tile: int = 560, max_tiles: int = 4, mean=CLIP_MEAN, std=CLIP_STD):
    """u8[h,w,3] -> (pixel_values f32[max_tiles,3,tile,tile], aspect_ratio_id, num_tiles, (tiles_h, tiles_w)).

    :483-541: resize into the optimal canvas (:431-481), zero-pad to the canvas (:392-429), rescale,
    normalise, split into tiles row-major (:39-49), pad the tile axis with zeros (:84-133);
    aspect_ratio_id = 1 + index of (tiles_h, tiles_w) in the supported list (:136-164)."""
    h, w = img.shape[:2]
    ch, cw = optimal_tiled_canvas(h, w, max_tiles, tile)
    th, tw = ch // tile, cw // tile
    new_h, new_w = fit_to_canvas_general(h, w, ch, cw, tile)
    small = pil_bilinear_resize_u8(img, new_h, new_w)
    canvas = np.zeros((ch, cw, 3), dtype=np.uint8)
    canvas[:new_h, :new_w] = small
    lut = normalise_lut(mean, std)
    full = np.stack([lut[c][canvas[:, :, c]] for c in range(3)])  # [3, ch, cw]
    tiles = full.reshape(3, th, tile, tw, tile).transpose(1, 3, 0, 2, 4).reshape(th * tw, 3, tile, tile)
    out = np.zeros((max_tiles, 3, tile, tile), dtype=np.float32)
    out[: th * tw] = tiles
    return out, supported_aspect_ratios(max_tiles).index((th, tw)) + 1, th * tw, (th, tw)
