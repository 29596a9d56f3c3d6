"""Where does the recogniser's time go on c5 pages (47 lines of 1.4-3.9 k px each)?  usage: c5_ocr_stages.py [pages_per_call]"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_ocr_amd import netspec, synth
from pero_ocr_amd.core.crop_engine import EngineLineCropper
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR
from pero_ocr_amd.ocr_engine import line_ocr_engine as loe

class Dev:
    type, index = "cuda", 0

ppb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
meta, spec, weights = bench.fixture_model("c2")
temp = float(os.environ.get("POCR_HEAD_TEMP", "1"))
weights = dict(weights)
weights["head.weight"] = weights["head.weight"] * np.float32(temp)
weights["head.bias"] = weights["head.bias"] * np.float32(temp)
tmp = tempfile.mkdtemp()
netspec.save_blob(os.path.join(tmp, "weights.pocrw"), spec, weights)
json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "weights.pocrw",
           "characters": meta["characters"][:-1], "net_name": "bench"}, open(os.path.join(tmp, "ocr.json"), "w"))
engine = PytorchEngineLineOCR(os.path.join(tmp, "ocr.json"), Dev(), batch_size=8)
crop = EngineLineCropper(line_height=spec.height, poly=2)
lines = []
for k in range(ppb):
    page = synth.make_page(900 + k, 3072, 4096)
    boxes = synth.page_line_boxes(900 + k, 3072, 4096)
    lines += crop.crop_lines(page, [(np.array([[x0, y0 + 30], [x0 + wd // 2, y0 + 30], [x0 + wd, y0 + 30]]), [30, 10]) for x0, y0, wd in boxes])
chunks = loe.plan_chunks([l.shape[1] for l in lines], engine.max_input_horizontal_pixels, engine.line_padding_px)
launches = loe.plan_launches(chunks, loe.launch_target(engine))
print("lines", len(lines), "chunks", len(chunks), "launches", [(len(l.line_ids), max(l.w_pads), sum(l.w_pads)) for l in launches])
import contextlib
for mode, kw in (("sparse+conf", dict(sparse_logits=True)), ("dense", dict(sparse_logits=False)), ("no logits", dict(no_logits=True))):
    try:
        with contextlib.redirect_stdout(sys.stderr):
            engine.process_lines(lines, **kw)
            t0 = time.perf_counter()
            for _ in range(5):
                engine.process_lines(lines, **kw)
            dt = (time.perf_counter() - t0) / 5
        print(mode, "ms per call", round(1e3 * dt, 2), "per page", round(1e3 * dt / ppb, 2))
    except TypeError as e:
        print(mode, "n/a", e)
with contextlib.redirect_stdout(sys.stderr):
    _t, lg, _c = engine.process_lines(lines, sparse_logits=True)
print("head temperature", temp, "frames", sum(m.shape[0] for m in lg), "nnz per frame", round(sum(m.nnz for m in lg) / sum(m.shape[0] for m in lg), 2),
      "distinct strings", len(set(_t)), "mean len", np.mean([len(t) for t in _t]))
engine.model.set_profiling(True)
with contextlib.redirect_stdout(sys.stderr):
    engine.process_lines(lines)
for sl in range(engine.model.num_slots):
    print("slot", sl, {k: round(v, 2) for k, v in engine.model.slot_stage_ms(sl).items()})
if os.environ.get("POCR_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(sys.stderr):
        pr.enable()
        for _ in range(3):
            engine.process_lines(lines, sparse_logits=True)
        pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
