import sys, os, json, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pero_ocr_amd import synth
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR
from pero_ocr_amd.ocr_engine import line_ocr_engine
class Dev: type, index = "cuda", 0
chars = synth.make_charset(231)
td = tempfile.mkdtemp(); path = os.path.join(td, "ocr.json")
json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent.pocrw", "characters": chars, "net_name": "b", "net": {"weight_seed": 20260929}}, open(path, "w"))
eng = PytorchEngineLineOCR(path, Dev(), batch_size=274)
base = synth.make_crops(305, [512]*256)
lines = [base[i % 256] for i in range(1024)]
eng.process_lines(lines[:300], no_logits=True)
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
for n in ("_pack_chunk", "_submit_chunk", "_collect_chunk"): wrap(eng, n)
for n in ("slot_stage_lines", "slot_launch", "slot_launch_sparse", "slot_collect", "slot_collect_sparse"): wrap(eng.model, n)
for kw in (dict(no_logits=True), dict()):
    T.clear(); t0 = time.perf_counter(); eng.process_lines(lines, **kw); tot = time.perf_counter() - t0
    print(kw, "total %.1f ms" % (tot*1e3), {k: round(v*1e3, 1) for k, v in T.items()})
