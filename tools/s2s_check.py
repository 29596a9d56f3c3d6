"""The s2s bench configuration (tools/s2s_bench.py) on a few lines: HIP engine vs the oracle (texts, logits)."""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd import netspec, synth
from pero_ocr_amd.ocr_engine import transformer_ocr_engine as tengine
from oracle import s2s_oracle
import torch
chars = synth.make_charset(231)
net = {"dim_model": 512, "dim_ff": 2048, "heads": 8, "encoder_layers": 2, "decoder_layers": 3, "conv_subsampling": [8, 4]}
td = tempfile.mkdtemp()
path = os.path.join(td, "ocr.json")
json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent", "characters": chars, "net_name": net,
           "max_line_width": 1024, "net": {"weight_seed": 20261002, "boundary_bias": 18.0}}, open(path, "w"))
eng = tengine.TransformerEngineLineOCR(path, torch.device("cuda:0"), batch_size=4)
crops = synth.make_crops(602, [512] * 6, 40)
got_t, got_l, _ = eng.process_lines(crops, sparse_logits=False)
spec = eng.net_spec if hasattr(eng, "net_spec") else None
print("spec", spec)
weights = netspec.generate_weights(spec, 20261002, boundary_bias=18.0)
model = s2s_oracle.OracleS2S(spec, weights)
want_t, want_l, _c, _ = s2s_oracle.process_lines(model, crops, eng.characters, 40, 480 * 4, 1024)
print("lens engine", [len(t) for t in got_t], "oracle", [len(t) for t in want_t])
print("texts equal", got_t == want_t)
for a, b in zip(got_l, want_l):
    n = min(len(a), len(b))
    print("max |dlogit| over common steps", float(np.max(np.abs(np.asarray(a)[:n] - np.asarray(b)[:n]))) if n else None, "steps", len(a), len(b))
