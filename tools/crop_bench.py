"""Line cropper on one page: 4k x 3k uint8 page, n lines of ~1500 px at height 40.
usage: python tools/crop_bench.py [n_lines] [reps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd.core.crop_engine import EngineLineCropper  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rng = np.random.RandomState(0)
    page = rng.randint(0, 256, size=(3000, 4000, 3)).astype(np.uint8)
    lines = []
    for i in range(n):
        y = 60 + (i * 2800) // n
        x0 = int(rng.randint(50, 400))
        pts = [[x0 + k * 500, y + int(rng.randint(-6, 7))] for k in range(4)]
        lines.append((np.array(pts), [28, 12]))
    eng = EngineLineCropper(line_height=40)
    eng.crop_lines(page, lines[:4])
    eng.crop_lines(page, lines)
    t0 = time.perf_counter()
    for _ in range(reps):
        specs = [eng.line_spec(b, h, 40) for b, h in lines]
    t_spec = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.set_page(page)
        eng._cropper.wait_page()
    t_up = (time.perf_counter() - t0) / reps
    res = {}
    for name, kw in (("copy", dict(copy=True)), ("views", dict(copy=False))):
        t0 = time.perf_counter()
        for _ in range(reps):
            crops = eng.crop_lines(page, lines, **kw)
        res[name] = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        crops = eng.crop_lines(None, lines, copy=False)
    t_res = (time.perf_counter() - t0) / reps
    px = sum(c.shape[0] * c.shape[1] for c in crops)
    print(json.dumps({"lines": n, "crop_pixels": px, "mean_width": round(px / 40 / n, 1),
                      "host_line_specs_ms": round(1e3 * t_spec, 3), "page_upload_alone_ms": round(1e3 * t_up, 3),
                      "crop_lines_ms_page_upload_to_numpy_copies": round(1e3 * res["copy"], 3),
                      "lines_per_s": round(n / res["copy"], 1),
                      "crop_lines_ms_views_of_pinned": round(1e3 * res["views"], 3), "lines_per_s_views": round(n / res["views"], 1),
                      "crop_lines_ms_resident_page": round(1e3 * t_res, 3), "lines_per_s_resident_page": round(n / t_res, 1),
                      "gpu_ms_measure_to_crops_in_pinned": round(eng._cropper.last_ms(), 3)}))


if __name__ == "__main__":
    main()
