"""Seeded synthetic text-line crops and charsets (there is no network for real
data).  Crops follow the reference's input contract (PageOCR hands
`uint8 [line_px_height, w, 3]` BGR arrays to process_lines,
pero_ocr/document_ocr/page_parser.py:418-423): light background, dark glyph-like
strokes, three identical channels ("grayscale" lines of BASELINE.json's configs).

Only integer hashing and IEEE +,-,*,/ and comparisons are used, so the same
(seed, index, width) gives the same bytes on every machine.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .netspec import splitmix64, uniform01


def make_crop(seed: int, index: int, width: int, height: int = 40) -> np.ndarray:
    """One `uint8 [height, width, 3]` crop."""
    stream = 0x5EED0000 + index
    # per-pixel background noise
    noise = uniform01(seed, stream, height * width).reshape(height, width)
    img = 215.0 + 30.0 * noise
    # glyph-like strokes: one pseudo-glyph every ~14 px, 2-4 segments each
    n_glyph = max(1, width // 14)
    par = uniform01(seed, stream ^ 0xABCDEF, n_glyph * 4 * 6).reshape(n_glyph, 4, 6)
    for g in range(n_glyph):
        gx0 = 2.0 + g * 14.0
        nseg = 2 + int(par[g, 0, 5] * 3.0)
        for s in range(min(nseg, 4)):
            p = par[g, s]
            x0 = gx0 + p[0] * 10.0
            y0 = 6.0 + p[1] * (height - 12.0)
            x1 = gx0 + p[2] * 10.0
            y1 = 6.0 + p[3] * (height - 12.0)
            thick = 0.9 + p[4] * 1.3
            # the stroke inks only pixels closer than sqrt(thick^2 + 1) < 2.5 px to the segment: evaluate a window
            # of +-4 columns around it (outside, ink is exactly 0 and the blend leaves the pixel bit-unchanged)
            c0 = max(0, int(min(x0, x1)) - 4)
            c1 = min(width, int(max(x0, x1)) + 6)
            if c1 <= c0:
                continue
            yy, xx = np.mgrid[0:height, c0:c1].astype(np.float64)
            dx, dy = x1 - x0, y1 - y0
            den = dx * dx + dy * dy + 1e-6
            t = ((xx - x0) * dx + (yy - y0) * dy) / den
            t = np.minimum(1.0, np.maximum(0.0, t))
            ex, ey = xx - (x0 + t * dx), yy - (y0 + t * dy)
            d2 = ex * ex + ey * ey
            ink = np.minimum(1.0, np.maximum(0.0, (thick * thick + 1.0 - d2) / (2.0 * thick)))
            img[:, c0:c1] = img[:, c0:c1] * (1.0 - ink) + (25.0 + 30.0 * p[5]) * ink
    g8 = np.minimum(255.0, np.maximum(0.0, np.floor(img + 0.5))).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g8[:, :, None], 3, axis=2))


def make_crops(seed: int, widths: Sequence[int], height: int = 40, indices: Sequence[int] = None) -> List[np.ndarray]:
    """Line i = make_crop(seed, indices[i], widths[i]); indices default to 0..n-1.  Fixtures whose lines were
    picked one by one (tests/golden/*.json "crop_indices") pass the picked indices."""
    if indices is None:
        indices = range(len(widths))
    return [make_crop(seed, int(k), int(w), height) for k, w in zip(indices, widths)]


def page_line_boxes(seed: int, height: int, width: int, line_height: int = 40, n_lines: int = None):
    """(x0, y0, width) of the text lines make_page pastes: the ground truth a layout post-processing stub can hand to
    the line cropper (baseline at 3/4 of the line height)."""
    pitch = int(line_height * 1.6)
    rows = max(1, (height - line_height) // pitch)
    n = rows if n_lines is None else min(n_lines, rows)
    u = uniform01(seed, 0x9A6F, 2 * n)
    boxes = []
    for k in range(n):
        wl = int((0.35 + 0.6 * u[2 * k]) * width)
        wl = max(8, min(wl, width - 8))
        x0 = int(u[2 * k + 1] * (width - wl))
        y0 = line_height // 2 + k * pitch
        if y0 + line_height > height:
            break
        boxes.append((x0, y0, wl))
    return boxes


def make_page(seed: int, height: int, width: int, line_height: int = 40, n_lines: int = None) -> np.ndarray:
    """A synthetic page `uint8 [height, width, 3]`: light noisy background with text lines (make_crop) pasted at seeded
    positions - the input of the layout network and of the line cropper."""
    noise = uniform01(seed, 0x9A6E, height * width).reshape(height, width)
    g = np.floor(225.0 + 25.0 * noise + 0.5).astype(np.uint8)
    page = np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))
    for k, (x0, y0, wl) in enumerate(page_line_boxes(seed, height, width, line_height, n_lines)):
        page[y0:y0 + line_height, x0:x0 + wl] = make_crop(seed + 17, k, wl, line_height)
    return page


def make_widths(seed: int, n: int, lo: int = 128, hi: int = 1024) -> List[int]:
    """n widths uniform in [lo, hi] (BASELINE config 3's seeded width distribution)."""
    u = uniform01(seed, 0x71D7, n)
    return [int(lo + np.floor(x * (hi - lo + 1))) for x in u]


def make_charset(n_symbols: int) -> List[str]:
    """A fixed printable charset with n_symbols entries (blank NOT included; the
    engine appends the blank placeholder itself, pytorch_ocr_engine.py:42).
    Latin + Czech diacritics + digits + punctuation first, then further
    Latin-Extended code points to fill up."""
    base = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
                " .,;:!?-()[]\"'/&%+=*§")
    base += list("áčďéěíňóřšťúůýžÁČĎÉĚÍŇÓŘŠŤÚŮÝŽäöüÄÖÜß")
    seen, out = set(), []
    for ch in base:
        if ch not in seen:
            seen.add(ch)
            out.append(ch)
    cp = 0x0100
    while len(out) < n_symbols:
        ch = chr(cp)
        if ch not in seen:
            seen.add(ch)
            out.append(ch)
        cp += 1
    return out[:n_symbols]


def random_u8_batch(seed: int, n: int, height: int, width: int) -> np.ndarray:
    """White-noise `uint8 [n, height, width, 3]` (edge-case / stress tests only)."""
    h = splitmix64(np.arange(n * height * width * 3, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x10001))
    return (h >> np.uint64(56)).astype(np.uint8).reshape(n, height, width, 3)
