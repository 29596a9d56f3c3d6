#!/usr/bin/env python3
"""Per-stage GPU times of ONE chunk run alone (blocking call, nothing overlapped).
Usage: python tools/stage_times.py [n_lines=256] [width=512] [arch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pero_ocr_amd import _native, netspec, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = int(sys.argv[2]) if len(sys.argv) > 2 else 512
arch = sys.argv[3] if len(sys.argv) > 3 else netspec.ARCH
spec = netspec.NetSpec(num_classes=232, arch=arch)
eng = _native.NativeEngine(spec, netspec.pack_weights(spec, netspec.generate_weights(spec, 1)), 0)
crops = synth.make_crops(1, [w] * min(n, 64))
pool = np.concatenate([crops[i % len(crops)].reshape(-1) for i in range(n)])
eng.stage_lines(pool, np.arange(n, dtype=np.int64) * (40 * w * 3), np.full(n, w, np.int32), w + 64, 32)
eng.run_staged(False, False)
eng.fallback_ready(wait=True)          # (the range guard's second engine is built on a thread behind pocr_create: its uploads and allocations
                                       #  would sit in the first timed runs)
eng.set_profiling(True)
runs = {}
for _ in range(7):
    eng.run_staged(False, False)
    for k, v in eng.last_stage_ms().items():
        runs.setdefault(k, []).append(v)
# the median of seven runs (one run in a few hundred shows a stage several ms long - something else on the box; a mean would carry it)
print({k: round(float(np.median(v)), 3) for k, v in runs.items()})
worst = {k: round(float(np.max(v)), 3) for k, v in runs.items() if np.max(v) > 1.5 * np.median(v) and np.max(v) > 0.1}
if worst:
    print("# slowest single run of a stage, where it is more than 1.5x the median:", worst)
