#!/usr/bin/env python3
"""End-to-end throughput of PytorchEngineLineOCR.process_lines (host crops in -> Python results out,
PCIe included) in its three output modes, on a seeded page stream.
Usage: python tools/e2e_bench.py [--lines 2048] [--width 512 | --ragged] [--batch-size 274]"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd import synth  # noqa: E402
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR  # noqa: E402


class Dev:
    type, index = "cuda", 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=2048)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--batch-size", type=int, default=274)
    ap.add_argument("--arch", default="vgg_blstm_ctc")
    a = ap.parse_args()
    chars = synth.make_charset(231)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ocr.json")
        json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent.pocrw", "characters": chars,
                   "net_name": "bench", "net": {"arch": a.arch, "weight_seed": 20260929}}, open(path, "w"))
        eng = PytorchEngineLineOCR(path, Dev(), batch_size=a.batch_size)
    widths = synth.make_widths(5, a.lines) if a.ragged else [a.width] * a.lines
    base = synth.make_crops(305, widths[:256])
    lines = [base[i % len(base)] if not a.ragged else None for i in range(a.lines)]
    if a.ragged:
        lines = synth.make_crops(305, widths)
    eng.process_lines(lines[:300], no_logits=True)          # warm-up
    out = {"lines": a.lines, "ragged": a.ragged, "width": None if a.ragged else a.width, "arch": a.arch}
    for name, kw in (("no_logits", dict(no_logits=True)), ("sparse_logits (default)", dict()),
                     ("dense_logits", dict(sparse_logits=False))):
        t0 = time.perf_counter()
        texts, logits, coords = eng.process_lines(lines, **kw)
        dt = time.perf_counter() - t0
        out[name] = round(a.lines / dt, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
