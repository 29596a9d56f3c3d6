"""Line cropper on the GPU: counterpart of the reference's EngineLineCropper
(pero_ocr/core/crop_engine.py:8-30, 54-111, 146-163) - same class name, constructor and `crop` contract.

`crop_lines` crops all lines of a page in three launches with the page resident in HBM (`pocr_cropper_*`,
csrc/crop.hpp + crop_host.hpp).  The host keeps the per-LINE scalars of get_crop_inputs (:54-72: integer baseline,
rotation, and the interpolant - scipy's cubic B-spline through a lean constructor that calls the same collocation and
LAPACK routines interp1d does, or np.polyfit); everything per column (walking the interpolated baseline, arc length,
resampling, normals: :73-89) and per pixel (:90-99, 146-163) runs on the device in float64, bit-identical to the
numpy / scipy sequence.  `line_curves` / `get_crop_inputs` are the host statement of the same mathematics
(return_forward_mapping, parity tests, and the older `pocr_crop_curves` entry point).

`return_mapping` (reverse mapping for blend_in, :113-145) is not built: nothing in the reference calls it.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import interpolate

from .. import _native

try:                                                   # the routines scipy.interpolate.make_interp_spline itself calls
    from scipy.interpolate import _dierckx as _sp_dierckx
    from scipy.linalg.lapack import dgbsv as _sp_dgbsv
except Exception:                                      # another scipy: interp1d does the work (slower, same numbers)
    _sp_dierckx = _sp_dgbsv = None

_LEAN_OK: Optional[bool] = None


def _lean_cubic(x: np.ndarray, y: np.ndarray):
    """Knots / coefficients of scipy.interpolate.interp1d(x, y, kind="cubic")._spline without interp1d's ~100 us of
    argument handling: the same steps with the same routines (mergesort, not-a-knot knot vector, `_coloc`, LAPACK gbsv
    - scipy/interpolate/_bsplines.py make_interp_spline), so the same bits.  Returns None when the input is not the
    plain case (fewer than 4 points, repeated or non-finite abscissae, singular system): the caller then lets interp1d
    itself decide (it raises, and the reference falls back to a straight line)."""
    n = x.size
    if n < 4 or not (np.isfinite(x).all() and np.isfinite(y).all()):
        return None
    ind = np.argsort(x, kind="mergesort")
    x, y = x[ind], y[ind]
    if np.any(x[1:] == x[:-1]):
        return None
    t = np.empty(n + 4)
    t[:4] = x[0]
    t[4:n] = x[2:-2]
    t[n:] = x[-1]
    ab = np.zeros((10, n), dtype=np.float64, order="F")
    _sp_dierckx._coloc(x, t, 3, ab.T, 0)
    _lu, _piv, c, info = _sp_dgbsv(3, 3, ab, y.reshape(-1, 1).copy(), overwrite_ab=True, overwrite_b=True)
    if info != 0:
        return None
    return t, c.ravel(), x[0], x[-1]


def cubic_interpolant(x: np.ndarray, y: np.ndarray):
    """(knots, coefficients, lo, hi) of interp1d(x, y, kind="cubic"); raises what interp1d raises."""
    global _LEAN_OK
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if _LEAN_OK is None:                               # once: the lean constructor must reproduce interp1d bit for bit
        _LEAN_OK = False
        if _sp_dierckx is not None and hasattr(_sp_dierckx, "_coloc"):
            px = np.array([3.0, 410.25, 977.5, 1502.0, 2101.75, 2600.1])
            py = np.array([7.0, 1.5, -3.25, 4.0, 2.0, -6.5])
            try:
                ref = interpolate.interp1d(px, py, kind="cubic")._spline
                got = _lean_cubic(px, py)
                _LEAN_OK = got is not None and np.array_equal(got[0], ref.t) and np.array_equal(got[1], ref.c.ravel())
            except Exception:
                _LEAN_OK = False
    if _LEAN_OK:
        got = _lean_cubic(x, y)
        if got is not None:
            return got
    f = interpolate.interp1d(x, y, kind="cubic")
    return f._spline.t, np.ascontiguousarray(f._spline.c).ravel(), f.x[0], f.x[-1]


def _poly1d_coeffs(c) -> np.ndarray:
    """np.poly1d(c).coeffs without the object (28 us): leading zeros trimmed, [0.] when nothing is left."""
    c = np.trim_zeros(np.atleast_1d(c), "f")
    return c if c.size else np.zeros(1)


class EngineLineCropper:
    def __init__(self, correct_slant=False, line_height=32, poly=0, scale=1, blend_border=4, device_id: int = 0):
        self.correct_slant = correct_slant
        self.line_height = line_height
        self.poly = poly
        self.scale = scale
        self.blend_border = blend_border
        self.device_id = device_id
        self._cropper: Optional[_native.NativeCropper] = None

    # ---- host part of the resident cropper: the per-line scalars -----------------------------------------------------
    def line_spec(self, baseline, line_heights, target_height):
        """What crop_engine.py:54-72 computes per line, and the extent of :73: (fields of pocr_crop_spec, knots,
        coefficients).  Raises what the reference's statements raise (the caller turns that into the fallback crop)."""
        above, below = line_heights[0] * self.scale, line_heights[1] * self.scale
        p = np.asarray(baseline).copy().astype(int)
        alpha = math.atan2(p[-1, 1] - p[0, 1], p[-1, 0] - p[0, 0])
        R = np.array([[np.cos(alpha), np.sin(alpha)], [-np.sin(alpha), np.cos(alpha)]])
        p = np.dot(p, np.linalg.inv(R))
        knots = None
        lo, hi = -np.inf, np.inf
        if self.poly:
            coefs = _poly1d_coeffs(np.polyfit(p[:, 0], p[:, 1], self.poly if p.shape[0] > 2 else 1))
        else:
            try:
                p[-1, 0] += 0.1
                knots, coefs, lo, hi = cubic_interpolant(p[:, 0], p[:, 1])
            except Exception:
                coefs = _poly1d_coeffs(np.polyfit(p[:, 0], p[:, 1], 1))
        x_min, x_max = p[:, 0].min(), p[:, 0].max()
        n_x = max(0, int(math.ceil(x_max - x_min)))       # len(np.arange(x_min, x_max))
        zoom = target_height / (above + below)
        return (x_min, x_max, lo, hi, zoom, above, below, R.reshape(-1), 0 if knots is not None else 1, n_x), knots, np.asarray(coefs, dtype=np.float64)

    # ---- host part: the line's 1-D curves --------------------------------------------------------------------
    def line_curves(self, baseline, line_heights, target_height):
        """The per-column half of get_crop_inputs (crop_engine.py:54-89): the baseline is rotated onto the x axis,
        interpolated (cubic spline, or a polynomial of degree `poly`) and sampled at `target_height / (up + down)`
        columns per unit of arc length - with the reference's own arc-length inversion (:101-111), which reduces to a
        straight interpolation between the first and last x.  Returns (curves float64 [4, w] = base_x, base_y,
        normal_x, normal_y; rows float64 [target_height] = offsets along the normal; R = the 2x2 rotation)."""
        above, below = line_heights[0] * self.scale, line_heights[1] * self.scale
        p = np.asarray(baseline).copy().astype(int)
        alpha = math.atan2(p[-1, 1] - p[0, 1], p[-1, 0] - p[0, 0])
        R = np.array([[np.cos(alpha), np.sin(alpha)], [-np.sin(alpha), np.cos(alpha)]])
        p = np.dot(p, np.linalg.inv(R))
        if self.poly:
            f = np.poly1d(np.polyfit(p[:, 0], p[:, 1], self.poly if p.shape[0] > 2 else 1))
        else:
            try:
                p[-1, 0] += 0.1                      # keeps the spline defined at the right end (:69)
                f = interpolate.interp1d(p[:, 0], p[:, 1], kind="cubic")
            except Exception:                        # too few points for a cubic: straight line (:71-72)
                f = np.poly1d(np.polyfit(p[:, 0], p[:, 1], 1))
        x = np.arange(p[:, 0].min(), p[:, 0].max())
        y = f(x)
        arc = np.concatenate([np.zeros(1), np.cumsum(((x[:-1] - x[1:]) ** 2 + (y[:-1] - y[1:]) ** 2) ** 0.5)])
        zoom = target_height / (above + below)
        t = np.linspace(0, arc[-1], int(arc[-1] * zoom))
        bx = self.reverse_line_mapping(arc, t, x)
        by = f(bx)
        ddx = np.full_like(bx, 0.1)
        ddy = by - f(bx + 0.1)
        length = (ddx ** 2 + ddy ** 2) ** 0.5        # (not np.hypot: the grid must round like the reference's)
        curves = np.stack((bx, by, -ddy / length, ddx / length))
        return curves, np.linspace(-above, below, target_height), R

    def get_crop_inputs(self, baseline, line_heights, target_height) -> np.ndarray:
        """float32 [target_height, w, 2]: (x, y) in the page of every crop pixel (crop_engine.py:54-99): every column
        of line_curves extended along its normal from -up to +down and rotated back.  (crop_lines does this last,
        2-D step on the GPU; this host version serves return_forward_mapping and the parity tests.)"""
        curves, rows, R = self.line_curves(baseline, line_heights, target_height)
        bx, by, normal_x, normal_y = curves
        v = rows.reshape(-1, 1)
        grid = np.stack((normal_x.reshape(1, -1) * v + bx.reshape(1, -1), normal_y.reshape(1, -1) * v + by.reshape(1, -1)), axis=2)
        return np.dot(grid, R).astype(np.float32)

    @staticmethod
    def reverse_line_mapping(forward_mapping, sample_positions, sampled_values) -> np.ndarray:
        """The reference's loop (:101-111) never advances its search pointer (forward_mapping[0] = 0 is not greater
        than any arc length), so it interpolates every sample on the wrap-around pair (last, first); vectorised."""
        last, first = forward_mapping[-1], forward_mapping[0]
        d = first - last
        da = (sample_positions - last) / d
        return (1 - da) * sampled_values[-1] + da * sampled_values[0]

    # ---- device part ---------------------------------------------------------------------------------------
    def set_page(self, img: np.ndarray):
        """Starts the upload of a page; `crop_lines(None, lines)` then crops from the resident copy, any number of times."""
        if self._cropper is None:
            self._cropper = _native.NativeCropper(self.device_id)
        self._cropper.set_page(img)
        self._page_ndim = img.ndim
        self._page_channels = img.shape[2] if img.ndim == 3 else 1

    def crop_lines(self, img: Optional[np.ndarray], lines: Sequence[Tuple[object, Sequence[float]]], copy: bool = True,
                   want_grids: bool = False, resident: bool = False):
        """All lines of a page: upload (behind the host's per-line work), three launches, one download.
        lines: (baseline, heights) pairs.  img None: the page of the last set_page.  A line whose grid cannot be
        computed gets the reference's fallback crop: zeros [line_height, 32, C] (crop_engine.py:20-22).
        copy=False: the crops are views of a pinned buffer that the next call overwrites.
        resident=True (3-channel pages): the crops stay in HBM and come back as `_native.LazyCrop` objects - array-likes that
        the recogniser stages on the GPU directly and that turn into numpy arrays only when their pixels are looked at."""
        if img is not None:
            self.set_page(img)
        elif self._cropper is None:
            raise ValueError("crop_lines(None, ...) needs set_page first")
        n = len(lines)
        specs = np.zeros(n, dtype=_native.CROP_SPEC_DTYPE)
        knots: List[np.ndarray] = []
        coefs: List[np.ndarray] = []
        nk = nc = 0
        failed = np.zeros(n, dtype=bool)
        rows = []
        for i, (baseline, heights) in enumerate(lines):
            try:
                head, kn, cf = self.line_spec(baseline, heights, self.line_height)
            except Exception:
                failed[i] = True
                rows.append(None)
                continue
            rows.append((head, len(cf), nc, nk))
            coefs.append(cf)
            nc += len(cf)
            if kn is not None:
                knots.append(kn)
                nk += len(kn)
        good = [i for i in range(n) if not failed[i]]
        if good:
            g = specs[:len(good)]
            for k, i in enumerate(good):
                head, ncf, co, ko = rows[i]
                g[k] = head[:7] + (head[7], head[8], ncf, co, ko, head[9], 0)
        channels = self._page_channels
        out: List[Optional[np.ndarray]] = [None] * n
        grids: List[Optional[np.ndarray]] = [None] * n
        if good:
            widths, _ = self._cropper.measure(specs[:len(good)], np.concatenate(knots) if knots else np.zeros(0),
                                               np.concatenate(coefs))
            if resident and not want_grids and self._page_ndim == 3 and channels == 3:
                res = self._cropper.crop_resident(self.line_height, self.device_id)
            else:
                res = self._cropper.crop(self.line_height, copy=copy, want_grids=want_grids)
            for k, i in enumerate(good):
                out[i] = res[0][k]
                if want_grids:
                    grids[i] = res[2][k]
        else:
            self._cropper.wait_page()
        for i in range(n):
            if out[i] is None:
                # the reference's crop() catches every failure of get_crop_inputs / fast_remap (an empty grid raises in
                # np.amin, crop_engine.py:147; interp1d raises outside its domain) and returns the zero crop
                print("ERROR: line crop failed.", lines[i][1], lines[i][0])
                out[i] = np.zeros([self.line_height, 32, channels], dtype=np.uint8)
            if self._page_ndim == 2:
                out[i] = out[i][:, :, 0]
        return (out, grids) if want_grids else out

    def crop(self, img, baseline, heights, return_mapping=False, return_forward_mapping=False):
        """crop_engine.py:16-30.  return_mapping: (crop, reverse mapping, offset) for `blend_in`; return_forward_mapping:
        (crop, float32 [H, w, 2] sampling grid).  The reference computes the reverse mapping from `line_coords` even when the
        crop failed (then line_coords is unbound and it raises UnboundLocalError / NameError); here that case raises ValueError."""
        (line_crop,) = self.crop_lines(img, [(baseline, heights)])
        if return_mapping:
            try:
                line_coords = self.get_crop_inputs(baseline, heights, self.line_height)
            except Exception as exc:
                raise ValueError("return_mapping: the line has no sampling grid (the crop is the fallback crop)") from exc
            line_mapping, offset = self.reverse_xy_mapping(line_coords, img.shape)
            return line_crop, line_mapping, offset
        if return_forward_mapping:
            return line_crop, self.get_crop_inputs(baseline, heights, self.line_height)
        return line_crop

    # ---- writing a (changed) crop back into the page: crop_engine.py:32-52, 113-145 (host code, like the reference) ----
    @staticmethod
    def _resize4_linear(a: np.ndarray) -> np.ndarray:
        """cv2.resize(a, (0, 0), fx=4, fy=4, interpolation=cv2.INTER_LINEAR) for a float 2-D array, restated from OpenCV's
        resize (pixel centres aligned: source x = (i + 0.5) / 4 - 0.5; outside [0, n - 1] the edge sample is taken with
        weight 1; float32 weights, rows then columns).  PARITY UNPINNED like the other OpenCV halves (no cv2 in this image)."""
        a = np.asarray(a, dtype=np.float32)

        def taps(n):
            f = (np.arange(4 * n, dtype=np.float32) + np.float32(0.5)) * np.float32(0.25) - np.float32(0.5)
            i0 = np.floor(f).astype(np.int64)
            w = (f - i0.astype(np.float32)).astype(np.float32)
            lo = i0 < 0
            hi = i0 >= n - 1
            i0 = np.clip(i0, 0, max(n - 1, 0))
            w = np.where(lo | hi, np.float32(0), w)
            return i0, np.minimum(i0 + 1, n - 1), w
        y0, y1, wy = taps(a.shape[0])
        x0, x1, wx = taps(a.shape[1])
        rows = a[:, x0] * (np.float32(1) - wx)[None, :] + a[:, x1] * wx[None, :]
        return (rows[y0] * (np.float32(1) - wy)[:, None] + rows[y1] * wy[:, None]).astype(np.float32)

    def reverse_xy_mapping(self, forward_mapping, shape):
        """crop_engine.py:113-137: the sampling grid upsampled x4, every upsampled sample rounded to its page pixel, and for
        each page pixel of the line's bounding box the crop coordinates (x, y) of the LAST sample that landed on it
        (-1: none).  -> (float32 [h, w, 2], (ystart, xstart))."""
        fm = np.asarray(forward_mapping)
        y_mapping = np.round(np.clip(self._resize4_linear(fm[:, :, 1]), 0, shape[0] - 1)).astype(int)
        x_mapping = np.round(np.clip(self._resize4_linear(fm[:, :, 0]), 0, shape[1] - 1)).astype(int)
        ystart, ystop = int(np.amin(y_mapping)), int(np.amax(y_mapping)) + 1
        xstart, xstop = int(np.amin(x_mapping)), int(np.amax(x_mapping)) + 1
        y_map = self._resize4_linear(np.tile(np.arange(0, fm.shape[0]), (fm.shape[1], 1)).T.astype(np.float32))
        x_map = self._resize4_linear(np.tile(np.arange(0, fm.shape[1]), (fm.shape[0], 1)).astype(np.float32))
        reverse_mapping = np.ones((ystop - ystart, xstop - xstart, 2), dtype=np.float32) * -1
        # the reference's Python loop assigns sample after sample in row-major order: the last one wins, which is what
        # numpy's indexed assignment does for repeated indices
        dy, dx = (y_mapping - ystart).reshape(-1), (x_mapping - xstart).reshape(-1)
        reverse_mapping[dy, dx, 0] = x_map.reshape(-1)
        reverse_mapping[dy, dx, 1] = y_map.reshape(-1)
        return reverse_mapping, (ystart, xstart)

    def get_blend_mask(self, mapping):
        """crop_engine.py:139-145."""
        from scipy import ndimage
        b = self.blend_border
        mask = mapping[:, :, 0] > -1
        mask = np.pad(mask, ((b, b), (b, b)))
        mask = ndimage.uniform_filter(mask.astype(float), size=2 * b + 1)
        mask = mask[b:-b, b:-b]
        mask = 2 * np.clip(mask - 0.5, 0, 1)
        return mask[:, :, np.newaxis]

    @staticmethod
    def _remap_transparent_u8(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray, dst: np.ndarray) -> None:
        """cv2.remap(src, map_x, map_y, INTER_LINEAR, borderMode=BORDER_TRANSPARENT, dst=dst) for 8-bit images, in place:
        OpenCV's fixed-point bilinear (coordinates to 1/32 pixel, 15-bit weights, (sum + 2^14) >> 15 - the arithmetic of
        csrc/crop.hpp) for the destination pixels whose 2 x 2 source footprint lies inside `src`; all others keep dst."""
        h, w = src.shape[:2]
        sx = np.rint(map_x.astype(np.float32) * np.float32(32)).astype(np.int64)
        sy = np.rint(map_y.astype(np.float32) * np.float32(32)).astype(np.int64)
        ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
        ok = (ix >= 0) & (iy >= 0) & (ix < w - 1) & (iy < h - 1)
        if not ok.any():
            return
        ixo, iyo, fxo, fyo = ix[ok], iy[ok], fx[ok], fy[ok]
        s = src.astype(np.int64)
        ex = (lambda a: a[:, None]) if src.ndim == 3 else (lambda a: a)
        acc = (ex((32 - fxo) * (32 - fyo) * 32) * s[iyo, ixo] + ex(fxo * (32 - fyo) * 32) * s[iyo, ixo + 1] +
               ex((32 - fxo) * fyo * 32) * s[iyo + 1, ixo] + ex(fxo * fyo * 32) * s[iyo + 1, ixo + 1])
        dst[ok] = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)

    def blend_in(self, img, line_crop, mapping, offset):
        """crop_engine.py:32-52: writes `line_crop` back over the page region it was cropped from (in place, returns img)."""
        ystart, xstart = offset[0], offset[1]
        ystop, xstop = ystart + mapping.shape[0], xstart + mapping.shape[1]
        blended_img = img[ystart:ystop, xstart:xstop].copy()
        mask = self.get_blend_mask(mapping)
        self._remap_transparent_u8(np.asarray(line_crop), mapping[:, :, 0], mapping[:, :, 1], blended_img)
        blended_img = np.round((1 - mask) * img[ystart:ystop, xstart:xstop] + mask * blended_img).astype(np.uint8)
        img[ystart:ystop, xstart:xstop] = blended_img
        return img
