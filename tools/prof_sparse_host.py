"""Host-side timeline of process_lines (default sparse mode vs no_logits): wall time spent inside every
native call, per launch.  usage: python tools/prof_sparse_host.py"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd import _native, synth  # noqa: E402
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR  # noqa: E402


class Dev:
    type, index = "cuda", 0


log = []


def wrap(name):
    fn = getattr(_native.NativeEngine, name)

    def inner(self, *a, **k):
        t0 = time.perf_counter()
        r = fn(self, *a, **k)
        log.append((name, t0, time.perf_counter()))
        return r
    setattr(_native.NativeEngine, name, inner)


def timed_collect_sparse(self, slot):
    """slot_collect_sparse with its three phases timed: wait for the launch, allocate, copy."""
    import ctypes as C
    import numpy as np
    n, T, rows, _wl, want_argmax, uni = self._slot_shape[slot]
    t0 = time.perf_counter()
    total = C.c_int64(0)
    self._lib.pocr_slot_sparse_nnz(self._h, int(slot), C.byref(total))
    t1 = time.perf_counter()
    data = np.empty(max(1, total.value), dtype=np.float32)
    indices = np.empty(max(1, total.value), dtype=np.int32)
    indptr = np.empty((n, self.spec.num_classes + 1), dtype=np.int32)
    line_off = np.empty(n + 1, dtype=np.int64)
    _lg, amax, labels, lens = self._alloc_out(n, T, False, want_argmax, None if uni else rows)
    t2 = time.perf_counter()
    P = _native._ptr
    self._lib.pocr_slot_collect_sparse(self._h, int(slot), P(data, _native._f32p), P(indices, _native._i32p), P(indptr, _native._i32p),
                                       P(line_off, _native._i64p), P(amax, _native._i32p), P(labels, _native._i32p), P(lens, _native._i32p))
    t3 = time.perf_counter()
    phases.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), total.value))
    return data[:total.value], indices[:total.value], indptr, line_off, amax, labels, lens


phases = []
_native.NativeEngine.slot_collect_sparse = timed_collect_sparse
for nm in ("slot_stage_ragged", "slot_launch", "slot_launch_sparse", "slot_collect", "slot_collect_sparse"):
    wrap(nm)
chars = synth.make_charset(231)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "ocr.json")
    json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent.pocrw", "characters": chars,
               "net_name": "bench", "net": {"arch": "vgg_blstm_ctc", "weight_seed": 20260929}}, open(path, "w"))
    eng = PytorchEngineLineOCR(path, Dev(), batch_size=8)
base = synth.make_crops(305, [512] * 256)
lines = [base[i % 256] for i in range(2048)]
eng.process_lines(lines[:300])
for kw in (dict(no_logits=True), dict()):
    del log[:]
    t0 = time.perf_counter()
    eng.process_lines(lines, **kw)
    t1 = time.perf_counter()
    print(kw, f"{2048 / (t1 - t0):.0f} lines/s, total {1e3 * (t1 - t0):.1f} ms")
    if phases:
        print("   collect_sparse phases (wait ms, alloc ms, copy ms, nnz):", [tuple(round(v, 1) for v in p) for p in phases[-9:]])
    print("   " + "  ".join(f"{n.replace('slot_', '')}@{1e3 * (a - t0):.1f}+{1e3 * (b - a):.1f}" for n, a, b in log))
