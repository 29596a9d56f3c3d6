"""Line cropper on the GPU: counterpart of the reference's EngineLineCropper
(pero_ocr/core/crop_engine.py:8-30, 54-111, 146-163) - same class name, constructor and `crop` contract.

The sampling grid (a few hundred 1-D operations per line: rotate the baseline, fit it, walk it at the target
resolution, add the normals) stays on the host in numpy/scipy, exactly as the reference computes it; the
per-pixel work - extending every column along its normal, rotating back, and the bilinear remap of
height x width x 3 samples per line - runs in a HIP kernel (`pocr_crop_curves`, csrc/crop.hpp), for all lines of
a page in one call (`crop_lines`).

`return_mapping` (the reverse mapping used for blending crops back into the page, :113-145) is not built.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import interpolate

from .. import _native


class EngineLineCropper:
    def __init__(self, correct_slant=False, line_height=32, poly=0, scale=1, blend_border=4, device_id: int = 0):
        self.correct_slant = correct_slant
        self.line_height = line_height
        self.poly = poly
        self.scale = scale
        self.blend_border = blend_border
        self.device_id = device_id

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
    def crop_lines(self, img: np.ndarray, lines: Sequence[Tuple[object, Sequence[float]]]) -> List[np.ndarray]:
        """All lines of a page in one GPU call.  lines: (baseline, heights) pairs.  A line whose grid cannot be
        computed gets the reference's fallback crop: zeros [line_height, 32, C] (crop_engine.py:20-22)."""
        parts: List[Optional[tuple]] = []
        for baseline, heights in lines:
            try:
                part = self.line_curves(baseline, heights, self.line_height)
                if part[0].shape[1] == 0:
                    # a grid without columns (arc length * zoom < 1): the reference's fast_remap raises on np.amin of the
                    # empty grid (crop_engine.py:147) and crop() falls back to the zero crop like for any other failure
                    raise ValueError("empty sampling grid")
                parts.append(part)
            except Exception:
                print("ERROR: line crop failed.", heights, baseline)
                parts.append(None)
        channels = img.shape[2] if img.ndim == 3 else 1
        good = [p for p in parts if p is not None]
        crops = iter(_native.crop_curves(img, [p[0] for p in good], [p[1] for p in good], [p[2] for p in good],
                                         self.device_id)) if good else iter(())
        out = []
        for p in parts:
            if p is None:
                out.append(np.zeros([self.line_height, 32, channels], dtype=np.uint8))
            else:
                c = next(crops)
                out.append(c if img.ndim == 3 else c[:, :, 0])
        return out

    def crop(self, img, baseline, heights, return_mapping=False, return_forward_mapping=False):
        if return_mapping:
            raise NotImplementedError("return_mapping (reverse mapping for blend_in) is not built for MI355X")
        (line_crop,) = self.crop_lines(img, [(baseline, heights)])
        if return_forward_mapping:
            return line_crop, self.get_crop_inputs(baseline, heights, self.line_height)
        return line_crop
