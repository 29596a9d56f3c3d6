"""Line cropper on one page: 4k x 3k uint8 page, n lines of ~1500 px at height 40.
usage: python tools/crop_bench.py [n_lines]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd.core.crop_engine import EngineLineCropper  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
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
    t0 = time.perf_counter()
    grids = [eng.get_crop_inputs(b, h, 40) for b, h in lines]
    t1 = time.perf_counter()
    crops = eng.crop_lines(page, lines)
    t2 = time.perf_counter()
    px = sum(c.shape[0] * c.shape[1] for c in crops)
    print(json.dumps({"lines": n, "crop_pixels": px, "host_grid_ms": round(1e3 * (t1 - t0), 2),
                      "crop_lines_ms_incl_grids_and_page_upload": round(1e3 * (t2 - t1), 2),
                      "lines_per_s": round(n / (t2 - t1), 1), "mean_width": round(px / 40 / n, 1)}))


if __name__ == "__main__":
    main()
