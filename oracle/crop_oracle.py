"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Line cropper: restatement of
pero_ocr/core/crop_engine.py:16-30 (crop), :54-99 (get_crop_inputs), :101-111 (reverse_line_mapping),
:146-163 (fast_remap), i.e. baseline + heights -> sampling grid -> bilinear remap -> uint8 [H, w, 3].

PARITY STATUS - two halves:
  * crop_inputs (the sampling grid) is PINNED: oracle/gen_golden_crop.py compiles the reference's own
    get_crop_inputs / reverse_line_mapping out of its source file (they need only numpy / scipy / math) and
    stores their outputs in tests/golden/crop_coords.npz; tests/test_crop.py compares this restatement with them.
  * remap_bilinear_u8 is PARITY UNPINNED: the reference calls cv2.remap(..., INTER_LINEAR, BORDER_CONSTANT)
    and OpenCV (dependency "opencv-python", no version pinned in the reference's pyproject.toml:28) is not
    installed in this image, so no reference output could be generated.  The function restates OpenCV's
    published fixed-point algorithm for 8-bit images (modules/imgproc/src/imgwarp.cpp: remap -> RemapInvoker
    converts float maps to 1/32-pixel fixed point with cvRound; remapBilinear with the 32x32 table of
    15-bit weights; FixedPtCast<int, uchar, 15> rounds (sum + 2^14) >> 15): per output pixel
        sx = rint(x * 32), sy = rint(y * 32)           (round half to even, float32 product)
        ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31
        w00 = (32-fx)(32-fy)*32, w01 = fx(32-fy)*32, w10 = (32-fx)fy*32, w11 = fx*fy*32    (sum = 32768, exact)
        out = (w00*p[iy][ix] + w01*p[iy][ix+1] + w10*p[iy+1][ix] + w11*p[iy+1][ix+1] + 16384) >> 15
    with pixels outside the image read as 0 (BORDER_CONSTANT, borderValue 0).
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
from scipy import interpolate


def invert_arclength(cum_length: np.ndarray, targets: np.ndarray, xs: np.ndarray) -> np.ndarray:
    """crop_engine.py:101-111 as written: a search pointer k advances while cum_length[k] > target and the result is
    interpolated on the "segment" [k-1, k].  Arc lengths are >= 0 = cum_length[0], so the pointer never moves, k - 1
    is index -1 (the LAST sample) and the function is a straight-line interpolation between the first and the last
    x over the whole length.  That is the reference's behaviour and what the fixtures pin."""
    out = np.zeros_like(targets)
    k = 0
    for i, t in enumerate(targets):
        while cum_length[k] > t:
            k += 1
        seg = cum_length[k] - cum_length[k - 1]
        frac = (t - cum_length[k - 1]) / seg
        out[i] = (1 - frac) * xs[k - 1] + frac * xs[k]
    return out


def crop_inputs(baseline: Sequence, heights: Sequence[float], target_height: int, scale: float = 1, poly: int = 0) -> np.ndarray:
    """-> float32 [target_height, w, 2] source (x, y) of every crop pixel."""
    up, down = heights[0] * scale, heights[1] * scale
    pts = np.asarray(baseline).copy().astype(int)
    angle = math.atan2(pts[-1, 1] - pts[0, 1], pts[-1, 0] - pts[0, 0])
    rot = np.array([[np.cos(angle), np.sin(angle)], [-np.sin(angle), np.cos(angle)]])
    pts = np.dot(pts, np.linalg.inv(rot))                       # baseline rotated onto the x axis
    if poly:
        curve = np.poly1d(np.polyfit(pts[:, 0], pts[:, 1], poly if pts.shape[0] > 2 else 1))
    else:
        try:
            pts[-1, 0] += 0.1
            curve = interpolate.interp1d(pts[:, 0], pts[:, 1], kind="cubic")
        except Exception:
            curve = np.poly1d(np.polyfit(pts[:, 0], pts[:, 1], 1))
    xs = np.arange(pts[:, 0].min(), pts[:, 0].max())
    ys = curve(xs)
    seg = ((xs[:-1] - xs[1:]) ** 2 + (ys[:-1] - ys[1:]) ** 2) ** 0.5
    cum = np.concatenate([np.zeros(1), np.cumsum(seg)])
    zoom = target_height / (up + down)
    n_cols = int(cum[-1] * zoom)
    base_x = invert_arclength(cum, np.linspace(0, cum[-1], n_cols), xs)
    base_y = curve(base_x)
    dx = np.full_like(base_x, 0.1)
    dy = base_y - curve(base_x + 0.1)
    norm = (dx ** 2 + dy ** 2) ** 0.5
    nx, ny = -dy / norm, dx / norm
    offs = np.linspace(-up, down, target_height).reshape(-1, 1)
    gx = nx.reshape(1, -1) * offs + base_x.reshape(1, -1)
    gy = ny.reshape(1, -1) * offs + base_y.reshape(1, -1)
    return np.dot(np.stack((gx, gy), axis=2), rot).astype(np.float32)


def remap_bilinear_u8(img: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """uint8 [H, W, C], float32 maps [h, w] -> uint8 [h, w, C]; see the module docstring (PARITY UNPINNED)."""
    H, W = img.shape[:2]
    sx = np.rint(map_x.astype(np.float32) * np.float32(32)).astype(np.int64)
    sy = np.rint(map_y.astype(np.float32) * np.float32(32)).astype(np.int64)
    # cv::saturate_cast<short> of the integer parts
    ix = np.clip(sx >> 5, -32768, 32767)
    iy = np.clip(sy >> 5, -32768, 32767)
    fx, fy = sx & 31, sy & 31
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    padded = np.zeros((H + 2, W + 2) + img.shape[2:], dtype=np.int64)
    padded[1:-1, 1:-1] = img

    def tap(yy, xx):
        ok = (yy >= -1) & (yy <= H) & (xx >= -1) & (xx <= W)
        v = padded[np.clip(yy + 1, 0, H + 1), np.clip(xx + 1, 0, W + 1)]
        return v * ok[..., None] if img.ndim == 3 else v * ok
    ex = (lambda a: a[..., None]) if img.ndim == 3 else (lambda a: a)
    acc = ex(w[0]) * tap(iy, ix) + ex(w[1]) * tap(iy, ix + 1) + ex(w[2]) * tap(iy + 1, ix) + ex(w[3]) * tap(iy + 1, ix + 1)
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def crop(img: np.ndarray, baseline, heights, line_height: int = 32, scale: float = 1, poly: int = 0) -> np.ndarray:
    """EngineLineCropper.crop (crop_engine.py:16-30): any failure gives a zero crop of 32 columns."""
    try:
        coords = crop_inputs(baseline, heights, line_height, scale, poly)
        if coords.shape[1] == 0:        # fast_remap's np.amin(coords) raises on an empty grid (crop_engine.py:147)
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        return remap_bilinear_u8(img, coords[:, :, 0], coords[:, :, 1])
    except Exception:
        return np.zeros([line_height, 32, img.shape[2]], dtype=np.uint8)
