"""The reference's default call (sparse logits) on 2048 x 40x512 crops: one call at a time against a stream of calls (call k + 1 begun before call k is
ended), several trials each, peaked (head x 8) and flat head.  usage: python tools/prof_stream_calls.py [calls_per_stream=6] [trials=4]"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_ocr_amd import netspec, synth
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR

n_stream = int(sys.argv[1]) if len(sys.argv) > 1 else 6
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 4
meta, spec, weights = bench.fixture_model("c2")
crops = synth.make_crops(305, [512] * 256, spec.height)
big = [crops[i % 256] for i in range(2048)]
keep = []
order = (1.0, 8.0) if os.environ.get("FLAT_FIRST") else (8.0, 1.0)
for temp in order:
    w = dict(weights)
    w["head.weight"] = w["head.weight"] * np.float32(temp); w["head.bias"] = w["head.bias"] * np.float32(temp)
    tmp = tempfile.mkdtemp()
    netspec.save_blob(os.path.join(tmp, "w.pocrw"), spec, w)
    json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "w.pocrw", "characters": meta["characters"][:-1], "net_name": "b"},
              open(os.path.join(tmp, "ocr.json"), "w"))
    eng = PytorchEngineLineOCR(os.path.join(tmp, "ocr.json"), bench.Dev(0), batch_size=274)
    eng.model.fallback_ready(wait=True)
    eng.process_lines(big); eng.process_lines(big)
    eng.model.device_synchronize()
    for t in range(trials):
        single = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.process_lines(big); eng.model.device_synchronize(); single.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ticket = eng.process_lines_begin(big)
        for _ in range(n_stream - 1):
            nxt = eng.process_lines_begin(big)
            eng.process_lines_end(ticket)
            ticket = nxt
        eng.process_lines_end(ticket)
        eng.model.device_synchronize()
        st = (time.perf_counter() - t0) / n_stream
        t0 = time.perf_counter()
        for _ in range(n_stream):
            eng.process_lines(big, no_logits=True)
        eng.model.device_synchronize()
        nl = (time.perf_counter() - t0) / n_stream
        print(f"head x{temp:g} trial {t}: one call at a time {[round(1e3 * d, 1) for d in single]} ms = {2048 / sorted(single)[1]:.0f} lines/s; "
              f"stream of {n_stream} calls {1e3 * st:.1f} ms per call = {2048 / st:.0f} lines/s; no_logits calls one at a time {1e3 * nl:.1f} ms = {2048 / nl:.0f} lines/s", flush=True)
    if os.environ.get("KEEP_ENGINES"):
        keep.append(eng)
    del eng
