"""Soak of the self-attention-encoder network (its fused conv1+2 kernel runs on 8 x 16 tiles, three workgroups per CU): random calls of
1..400 lines of 1..2000 px through process_lines in all output modes; a repeated call must return exactly what its first pass returned.
usage: python tools/stress_sa.py [seconds]"""
import json, os, sys, tempfile, time, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_ocr_amd import netspec, synth
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
meta, spec, weights = bench.fixture_model("c4")
tmp = tempfile.mkdtemp()
netspec.save_blob(os.path.join(tmp, "w.pocrw"), spec, weights)
json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "w.pocrw", "characters": meta["characters"][:-1], "net_name": "sa"},
          open(os.path.join(tmp, "ocr.json"), "w"))
eng = PytorchEngineLineOCR(os.path.join(tmp, "ocr.json"), bench.Dev(0), batch_size=8)
rng = np.random.RandomState(1)
cases = []
for k in range(12):
    n = int(rng.randint(1, 401))
    widths = [int(w) for w in rng.choice([1, 9, 40, 64, 130, 257, 300, 512, 768, 1000, 1500, 2000], size=n)]
    cases.append((synth.make_crops(2000 + k, widths, spec.height), [dict(), dict(sparse_logits=False), dict(no_logits=True)][k % 3]))
first, calls, lines, t_end = {}, 0, 0, time.time() + seconds
with contextlib.redirect_stdout(sys.stderr):
    while time.time() < t_end:
        k = int(rng.randint(0, len(cases)))
        crops, kw = cases[k]
        t, l, c = eng.process_lines(crops, **kw)
        sig = (t, c, None if l[0] is None else [float(np.asarray(m.sum())) if hasattr(m, "nnz") else float(m.sum()) for m in l])
        if k in first:
            assert first[k] == sig, f"case {k}: a repeated call returned something else"
        first[k] = sig
        calls += 1; lines += len(crops)
print(json.dumps({"seconds": seconds, "calls": calls, "lines": lines, "distinct_cases": len(first), "result": "every repeat identical"}))
