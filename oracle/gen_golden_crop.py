"""ORACLE TOOLING — runs ONLY in the build container (needs /root/reference).

Line-cropper sampling grids: compiles the reference's OWN
EngineLineCropper.get_crop_inputs / reverse_line_mapping (pero_ocr/core/crop_engine.py:54-111) out of its source
file with `ast` (the module itself imports cv2, which is absent; these two methods need numpy / scipy / math only),
runs them on seeded baselines and stores the float32 grids in tests/golden/crop_coords.npz.
The remap step (cv2.remap) cannot be run here: see oracle/crop_oracle.py.
"""
from __future__ import annotations

import ast
import json
import math
import os
import sys

import numpy as np
from scipy import interpolate

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REFERENCE = "/root/reference"

CASES = [   # (name, baseline points, heights, line_height, scale, poly)
    ("straight", [[100, 200], [400, 200]], [30, 10], 40, 1, 0),
    ("slanted", [[50, 300], [220, 330], [400, 372]], [25, 9], 40, 1, 0),
    ("curved", [[10, 100], [90, 108], [180, 120], [260, 118], [350, 104], [430, 96]], [28, 12], 40, 1, 0),
    ("short2", [[500, 40], [530, 44]], [20, 8], 40, 1, 0),
    ("poly2", [[10, 100], [90, 108], [180, 120], [260, 118], [350, 104]], [28, 12], 48, 1, 2),
    ("scaled", [[60, 500], [200, 480], [380, 470], [520, 474]], [22, 7], 32, 1.25, 0),
    ("steep", [[300, 100], [340, 260], [372, 420]], [18, 6], 40, 1, 0),
]


def reference_methods():
    ns = {"np": np, "math": math, "interpolate": interpolate}
    tree = ast.parse(open(os.path.join(REFERENCE, "pero_ocr/core/crop_engine.py"), encoding="utf8").read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in ("get_crop_inputs", "reverse_line_mapping"):
            node.decorator_list = []                    # @jit(nopython=False, forceobj=True): plain Python semantics
            exec(compile(ast.Module(body=[node], type_ignores=[]), "crop_engine.py", "exec"), ns)

    class Cropper:                                      # the attributes the two methods read
        def __init__(self, scale, poly):
            self.scale, self.poly = scale, poly
        get_crop_inputs = ns["get_crop_inputs"]
        reverse_line_mapping = ns["reverse_line_mapping"]
    return Cropper


def main():
    Cropper = reference_methods()
    arrays, meta = {}, []
    for name, baseline, heights, line_height, scale, poly in CASES:
        coords = Cropper(scale, poly).get_crop_inputs(np.array(baseline), heights, line_height)
        assert coords.dtype == np.float32 and coords.shape[0] == line_height and coords.shape[2] == 2
        arrays[name] = coords
        meta.append({"name": name, "baseline": baseline, "heights": heights, "line_height": line_height, "scale": scale,
                     "poly": poly, "shape": list(coords.shape)})
        print(name, coords.shape, float(coords[..., 0].min()), float(coords[..., 0].max()))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "crop_coords.npz"), **arrays)
    with open(os.path.join(REPO, "tests", "golden", "crop_coords.json"), "w", encoding="utf8") as f:
        json.dump({"cases": meta, "numpy": np.__version__}, f, indent=0)


if __name__ == "__main__":
    main()
