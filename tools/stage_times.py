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
eng.set_profiling(True)
acc = {}
for _ in range(5):
    eng.run_staged(False, False)
    for k, v in eng.last_stage_ms().items():
        acc[k] = acc.get(k, 0) + v / 5
print({k: round(v, 3) for k, v in acc.items()})
