"""Where does the reference's DEFAULT call (process_lines with sparse logits) spend its host time on c2-shaped input?
usage: python tools/prof_default_call.py"""
import cProfile
import json
import os
import pstats
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pero_ocr_amd import netspec, synth  # noqa: E402
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR  # noqa: E402

meta, spec, weights = bench.fixture_model("c2")
weights = dict(weights)
weights["head.weight"] = weights["head.weight"] * np.float32(8)
weights["head.bias"] = weights["head.bias"] * np.float32(8)
tmp = tempfile.mkdtemp()
netspec.save_blob(os.path.join(tmp, "w.pocrw"), spec, weights)
json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "w.pocrw", "characters": meta["characters"][:-1], "net_name": "b"},
          open(os.path.join(tmp, "ocr.json"), "w"))
eng = PytorchEngineLineOCR(os.path.join(tmp, "ocr.json"), bench.Dev(0), batch_size=274)
crops = synth.make_crops(305, [512] * 256, spec.height)
big = [crops[i % 256] for i in range(2048)]
eng.process_lines(big[:512])
for kw in (dict(no_logits=True), dict()):
    t0 = time.perf_counter()
    eng.process_lines(big, **kw)
    print(kw, f"{2048 / (time.perf_counter() - t0):.0f} lines/s")
pr = cProfile.Profile()
pr.enable()
eng.process_lines(big)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
