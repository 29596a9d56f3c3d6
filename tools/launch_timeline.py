"""Host timeline of one process_lines call: when each launch was submitted / collected and how long the calls took.
usage: python tools/launch_timeline.py [taper-spec] [n_lines] [batch_size]"""
import json, os, sys, tempfile, time, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_ocr_amd import netspec, synth
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR

os.environ["POCR_LAUNCH_TAPER"] = sys.argv[1] if len(sys.argv) > 1 else "0"
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 274
kw = dict(no_logits=True) if os.environ.get("NO_LOGITS") else {}
meta, spec, weights = bench.fixture_model("c2")
weights = dict(weights)
temp = np.float32(os.environ.get("POCR_HEAD_TEMP", "8"))          # 8: peaked posteriors (4 entries per frame kept); 1: the seeded flat head (208 of 232)
weights["head.weight"] = weights["head.weight"] * temp; weights["head.bias"] = weights["head.bias"] * temp
tmp = tempfile.mkdtemp()
netspec.save_blob(os.path.join(tmp, "w.pocrw"), spec, weights)
json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "w.pocrw", "characters": meta["characters"][:-1], "net_name": "b"},
          open(os.path.join(tmp, "ocr.json"), "w"))
crops = synth.make_crops(305, [512] * 256, spec.height)
big = [crops[i % 256] for i in range(n_lines)]
if os.environ.get("MIX"):                                          # the c3 width mix (128..1024 px) instead of uniform 512-px lines
    big = synth.make_crops(77, synth.make_widths(33, n_lines), spec.height)
eng = PytorchEngineLineOCR(os.path.join(tmp, "ocr.json"), bench.Dev(0), batch_size=bs)
ev = []
sub, col = eng._submit_launch, eng._collect_launch
def submit(lines, launch, *a, **k):
    t0 = time.perf_counter(); h = sub(lines, launch, *a, **k); ev.append(("submit", len(launch.line_ids), t0, time.perf_counter())); return h
def collect(h):
    t0 = time.perf_counter(); r = col(h); ev.append(("collect", h[1], t0, time.perf_counter())); return r
eng._submit_launch, eng._collect_launch = submit, collect
with contextlib.redirect_stdout(sys.stderr):
    eng.process_lines(big, **kw); eng.process_lines(big, **kw)
    eng.model.device_synchronize()
    ev.clear()
    T0 = time.perf_counter()
    eng.process_lines(big, **kw)
    T1 = time.perf_counter()
print(f"call {1e3 * (T1 - T0):.2f} ms, taper {os.environ['POCR_LAUNCH_TAPER']}")
for kind, x, a, b in ev:
    print(f"  {kind:8s} {x:4d}  at {1e3 * (a - T0):7.2f}  took {1e3 * (b - a):6.2f} ms")
