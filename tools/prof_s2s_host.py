"""Host-side timeline of TransformerEngineLineOCR.process_lines: wall time inside the native calls, per launch.
usage: python tools/prof_s2s_host.py [n_lines]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd import _native, synth  # noqa: E402
from pero_ocr_amd.ocr_engine.transformer_ocr_engine import TransformerEngineLineOCR  # noqa: E402


class Dev:
    type, index = "cuda", 0


log = []
for nm in ("s2s_stage", "s2s_launch", "s2s_decode", "s2s_sparse"):
    def make(fn, nm=nm):
        def inner(self, *a, **k):
            t0 = time.perf_counter()
            r = fn(self, *a, **k)
            log.append((nm, t0, time.perf_counter()))
            return r
        return inner
    setattr(_native.NativeEngine, nm, make(getattr(_native.NativeEngine, nm)))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chars = synth.make_charset(231)
net = {"dim_model": 512, "dim_ff": 2048, "heads": 8, "encoder_layers": 2, "decoder_layers": 3, "conv_subsampling": [8, 4]}
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "ocr.json")
    json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent", "characters": chars, "net_name": net,
               "max_line_width": 1024, "net": {"weight_seed": 20261002, "boundary_bias": 20.0}}, open(path, "w"))
    eng = TransformerEngineLineOCR(path, Dev(), batch_size=4)
crops = synth.make_crops(602, [512] * n, 40)
eng.process_lines(crops[:300])
for kw in (dict(no_logits=True), dict()):
    for rep in range(2):
        del log[:]
        t0 = time.perf_counter()
        eng.process_lines(crops, **kw)
        t1 = time.perf_counter()
    print(kw, f"{n / (t1 - t0):.0f} lines/s, total {1e3 * (t1 - t0):.1f} ms")
    print("   " + "  ".join(f"{nm[4:]}@{1e3 * (a - t0):.1f}+{1e3 * (b - a):.1f}" for nm, a, b in log))
