#!/usr/bin/env python3
"""How accurate is the GPU path compared with the reference arithmetic itself?  Runs 48 lines of the
c2 workload through (a) the oracle in float64, (b) the oracle in float32 (= the reference's CPU
arithmetic), (c) the HIP engine, and prints max / rms logit errors of (b) and (c) against (a)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pero_ocr_amd import _native, netspec, synth
from oracle import engine_oracle, model_oracle
spec = netspec.NetSpec(num_classes=232)
w = netspec.generate_weights(spec, 20260929)
crops = synth.make_crops(305, [512] * 48)
batch = engine_oracle.assemble_batch(crops, list(range(48)), 40, 512, 480 * 274)
net = model_oracle.OracleNet(spec, w)
ref32 = model_oracle.forward_logits(net, batch)
net64 = model_oracle.OracleNet(spec, w).double()
with torch.no_grad():
    ref64 = net64((torch.from_numpy(batch).double() / 255.0).permute(0, 3, 1, 2)).numpy()
eng = _native.NativeEngine(spec, netspec.pack_weights(spec, w), 0)
gpu, amax, labels, lens = eng.run_batch(batch)
gpu = gpu.transpose(0, 2, 1)
for name, x in (("reference arithmetic (torch CPU fp32)", ref32), ("HIP engine (fp32 MFMA)", gpu)):
    d = np.abs(x.astype(np.float64) - ref64)
    print(f"{name:40s} vs fp64: max {d.max():.3e}  rms {np.sqrt((d ** 2).mean()):.3e}")
d = np.abs(gpu - ref32)
print(f"{'HIP engine vs reference arithmetic':40s}        : max {d.max():.3e}  rms {np.sqrt((d ** 2).mean()):.3e}")
print("argmax equal to fp64:", "ref32", int((ref32.argmax(1) != ref64.argmax(1)).sum()), "flips; gpu", int((gpu.argmax(1) != ref64.argmax(1)).sum()), "flips of", ref64.shape[0] * ref64.shape[2])
