"""Where a c3 pass spends its time (one GPU): wall time inside the host functions of the sharded page stream, and the GPU
stage times (HIP events) of every launch.  python tools/prof_c3.py"""
import sys, os, json, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from pero_ocr_amd import sharding, synth, netspec
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR


class Dev:
    type, index = "cuda", 0


meta, spec, weights = bench.fixture_model("c3")
tmp = tempfile.mkdtemp()
netspec.save_blob(os.path.join(tmp, "weights.pocrw"), spec, weights)
json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "weights.pocrw",
           "characters": meta["characters"][:-1], "net_name": "prof"}, open(os.path.join(tmp, "ocr.json"), "w", encoding="utf8"))
engine = PytorchEngineLineOCR(os.path.join(tmp, "ocr.json"), Dev(), batch_size=8)
lines = synth.make_crops(meta["crop_seed"], meta["widths"], spec.height, meta.get("crop_indices"))
sh = sharding.ShardedLineOCR(sharding.engine_recogniser(engine), engine.characters, engine.max_input_horizontal_pixels,
                             transport=sharding.LocalTransport())
sh.process_lines(lines, no_logits=True)
T, launches = {}, []


def wrap(obj, name, post=None):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        if post:
            post(a, r)
        return r
    setattr(obj, name, g)


m = engine.model
works = []
for n in ("_pack_lines", "_collect_launch"):
    if hasattr(engine, n):
        wrap(engine, n)
wrap(engine, "_submit_launch", post=lambda a, r: works.append((len(a[1].line_ids), a[1].work, max(a[1].w_pads), min(a[1].w_pads))))
for n in ("slot_stage_ragged", "slot_stage_lines", "slot_launch"):
    if hasattr(m, n):
        wrap(m, n)
wrap(m, "slot_collect", post=lambda a, r: launches.append(dict(m.slot_stage_ms(a[0]))))
m.set_profiling(True)
for rep in range(2):
    T.clear(); launches.clear(); works.clear()
    t0 = time.perf_counter(); sh.process_lines(lines, no_logits=True); tot = time.perf_counter() - t0
    print("pass %.1f ms; host wall inside: %s" % (tot * 1e3, {k: round(v * 1e3, 1) for k, v in T.items()}))
    print("launches %d; GPU total per launch (ms): %s" % (len(launches), [round(l.get("total", 0), 1) for l in launches]))
    conv = [round(sum(v for k, v in l.items() if k.startswith("conv") or k == "agg"), 1) for l in launches]
    print("  conv+agg per launch:", conv, "sum", round(sum(conv), 1), "; lstm per launch:", [round(l.get("lstm", 0), 1) for l in launches])
    if os.environ.get("POCR_PIPELINE_DEPTH") == "1":       # one launch at a time: the stage events are the launch's own
        print("  per launch alone: lines, padded columns, widest / narrowest W_pad, conv+agg ms, ns per column, per-layer ms")
        for (nl, w, wmax, wmin), l in zip(works, launches):
            c = sum(v for k, v in l.items() if k.startswith("conv") or k == "agg")
            print("   %4d %7d %5d %5d  %6.2f ms  %5.1f ns/col  %s" % (nl, w, wmax, wmin, c, 1e6 * c / w, {k: round(v, 2) for k, v in l.items() if k.startswith("conv") or k in ("agg", "lstm")}))
