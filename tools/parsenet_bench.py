#!/usr/bin/env python3
"""Layout network on one 4k x 3k page at downsample 4 (BASELINE config 5's first stage): GPU ms per page (HIP events:
down-sampling + 20 conv layers + head), wall ms incl. PCIe, achieved TFLOP/s.  Usage: python tools/parsenet_bench.py [H W ds reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pero_ocr_amd import _native, parsenet_spec as ps, synth
H = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
net = _native.NativeParseNet(ps.pack_weights(ps.generate_weights(1)), 0)
page = synth.make_page(2, H, W, n_lines=20)
h, w = net.out_shape(H, W, ds)
hp, wp = ps.padded_shape(h, w)
flops = 0.0
lvl = 0
for name, cin, cout, pool in ps.ENCODER:
    flops += 2.0 * (hp >> lvl) * (wp >> lvl) * cin * cout * 9
    lvl += 1 if pool == 2 else 0
for k, (name, cup, cskip, cout) in zip(range(5, -1, -1), ps.DECODER):
    flops += 2.0 * (hp >> k) * (wp >> k) * (cup + cskip) * cout * 9
flops += 2.0 * h * w * 64 * 5
net.get_maps(page, ds)
gpu, wall = [], []
for _ in range(reps):
    t0 = time.perf_counter(); net.get_maps(page, ds); wall.append(time.perf_counter() - t0); gpu.append(net.last_ms())
g, wl = float(np.median(gpu)), 1e3 * float(np.median(wall))
print(json.dumps({"page": [H, W], "downsample": ds, "net_input": [hp, wp], "gflop_per_page": round(flops / 1e9, 1),
                  "gpu_ms_per_page": round(g, 3), "wall_ms_per_page_incl_pcie": round(wl, 3), "achieved_tflops": round(flops / g / 1e9, 1),
                  "vs_fp32_mfma_peak_157.3": round(flops / g / 1e9 / 157.3, 3), "frac_of_bf16x3_ceiling_416.7": round(flops / g / 1e9 / 416.7, 3), "pages_per_s_wall": round(1e3 / wl, 1)}))
