"""ORACLE TOOLING — runs ONLY in the build container (needs /root/reference).

Line confidences: runs the reference's OWN
PageParser.compute_line_confidence / get_prob (pero_ocr/document_ocr/page_parser.py:485-496, 437-450)
and TextLine.get_dense_logits (pero_ocr/core/layout.py:65-68) on the sparse logits stored in the c1
fixture (which the reference engine produced) and writes the values to tests/golden/c1_confidence.json.

page_parser.py and layout.py cannot be imported here (cv2, lxml, shapely, ... are absent), so the three
function definitions are taken out of the reference's source files with `ast` and compiled on their own,
with numpy as their only global.  Nothing of the reference is copied into the repo: only the numbers.
"""
from __future__ import annotations

import ast
import json
import os
import sys

import numpy as np
from scipy import sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REFERENCE = "/root/reference"


def reference_functions():
    ns = {"np": np}

    def grab(path, names):
        tree = ast.parse(open(os.path.join(REFERENCE, path), encoding="utf8").read())
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name in names:
                node.decorator_list = []                       # compute_line_confidence is a @staticmethod
                mod = ast.Module(body=[node], type_ignores=[])
                exec(compile(mod, path, "exec"), ns)
    grab("pero_ocr/document_ocr/page_parser.py", {"get_prob", "compute_line_confidence"})
    grab("pero_ocr/core/layout.py", {"get_dense_logits"})
    return ns


class Line:                         # the two members compute_line_confidence touches
    def __init__(self, logits, get_dense):
        self.logits = logits
        self._get_dense = get_dense

    def get_dense_logits(self, zero_logit_value: int = -80):
        return self._get_dense(self, zero_logit_value)


def main():
    ns = reference_functions()
    meta = json.load(open(os.path.join(REPO, "tests", "golden", "c1.json"), encoding="utf8"))
    z = np.load(os.path.join(REPO, "tests", "golden", "c1.npz"))
    n = len(meta["widths"])
    C = len(meta["characters"])
    conf = []
    for i in range(n):
        T = int(z["shapes"][i][0])
        m = sparse.csc_matrix((z[f"csc_data_{i}"], z[f"csc_indices_{i}"], z[f"csc_indptr_{i}"]), shape=(T, C))
        conf.append(float(ns["compute_line_confidence"](Line(m, ns["get_dense_logits"]))))
    # a hand-made case with runs, a tie and an all-dropped frame
    lg = np.array([[5.0, 0.0, 0.0, 1.0], [4.0, 0.0, 3.5, 0.0], [0.0, 2.0, 0.0, 0.0], [0.0, 2.5, 0.0, 0.0],
                   [0.0, 0.0, 0.0, 0.0], [1.0, 0.0, 1.0, 0.0]], dtype=np.float32)
    hand = float(ns["compute_line_confidence"](Line(sparse.csc_matrix(lg), ns["get_dense_logits"])))
    out = {"source": "c1 sparse logits (reference engine run) -> reference compute_line_confidence",
           "confidence": conf, "hand_logits": lg.tolist(), "hand_confidence": hand, "numpy": np.__version__}
    with open(os.path.join(REPO, "tests", "golden", "c1_confidence.json"), "w", encoding="utf8") as f:
        json.dump(out, f, indent=0)
    print("confidences:", np.round(conf[:8], 6), "... hand case:", hand)


if __name__ == "__main__":
    main()
