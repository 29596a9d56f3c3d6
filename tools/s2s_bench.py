"""End-to-end throughput of the sequence-to-sequence engine on one MI355X:
TransformerEngineLineOCR.process_lines on synthetic crops / seeded weights.
usage: python tools/s2s_bench.py [n_lines] [width | 0 = ragged 128..2000] [batch_size] [dec_layers] [boundary_bias] [lines_per_launch]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd import synth  # noqa: E402
from pero_ocr_amd.ocr_engine.transformer_ocr_engine import TransformerEngineLineOCR  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    batch_size = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    dec_layers = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    bias = float(sys.argv[5]) if len(sys.argv) > 5 else 20.0
    import torch
    from pero_ocr_amd.ocr_engine import transformer_ocr_engine as te
    if len(sys.argv) > 6:
        te.LAUNCH_MAX_LINES = int(sys.argv[6])
        te.LAUNCH_MAX_COLUMNS = int(sys.argv[6]) * 1088
    chars = synth.make_charset(231)
    net = {"dim_model": 512, "dim_ff": 2048, "heads": 8, "encoder_layers": 2, "decoder_layers": dec_layers,
           "conv_subsampling": [8, 4]}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ocr.json")
        with open(path, "w", encoding="utf8") as f:
            json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent", "characters": chars,
                       "net_name": net, "max_line_width": 1024, "net": {"weight_seed": 20261002, "boundary_bias": bias}}, f)
        eng = TransformerEngineLineOCR(path, torch.device("cuda:0"), batch_size=batch_size)
    widths = synth.make_widths(602, n, 128, 2000) if width == 0 else [width] * n      # width 0: ragged 128..2000 px
    crops = synth.make_crops(602, widths, 40)
    eng.process_lines(crops, no_logits=True)                 # warm-up: the same call (every slot's buffers reach their size; a cold slot costs a timed pass seconds)
    out = {}
    for mode, kw in (("no_logits", dict(no_logits=True)), ("dense", dict(sparse_logits=False)), ("sparse", {})):
        t0 = time.perf_counter()
        texts, _l, _c = eng.process_lines(crops, **kw)
        dt = time.perf_counter() - t0
        out[mode] = round(n / dt, 1)
    lens = np.array([len(t) for t in texts])
    # (line, step) pairs the memory attention really evaluates: a line is skipped once its REFERENCE BATCH (batch_size consecutive lines
    # of the width-sorted order; equal widths: input order) has ended, and a batch runs max(len) + 1 steps (the boundary symbol's step)
    if width:
        groups = [lens[i:i + batch_size] for i in range(0, n, batch_size)]
        out["attention_line_steps_per_pass"] = int(sum(len(g) * (int(g.max()) + 1) for g in groups))
        out["passes"] = 4                                    # warm-up + the three timed modes: what a kernel trace of this command contains
    out.update(lines=n, width=width, batch_size=batch_size, dec_layers=dec_layers,
               mean_len=float(lens.mean()), share_at_limit=float((lens >= 272).mean()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
