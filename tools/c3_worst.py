#!/usr/bin/env python3
"""Where do a fixture's logit deviations come from?  HIP vs the reference rows and (when the fixture has them) vs the
float64 truth rows; prints the worst lines and the distribution."""
import os, sys, tempfile
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from conftest import Golden
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR
class Dev: type, index = "cuda", 0
g = Golden(sys.argv[1] if len(sys.argv) > 1 else "c3")
eng = PytorchEngineLineOCR(g.write_engine_json(tempfile.mkdtemp()), Dev(), batch_size=g.batch_size)
texts, logits, coords = eng.process_lines(g.crops(), sparse_logits=False)
flips = sum(int(np.sum(np.argmax(np.asarray(logits[i]), axis=1) != g.argmax(i))) for i in range(g.n))
e_ref, e_truth, r_truth = [], [], []
for i in range(g.n):
    got = np.asarray(logits[i])[g.sample_rows[i]]
    e_ref.append(np.abs(got - g.rows(i)).max(axis=1))
    t = g.rows64(i)
    if t is not None:
        e_truth.append(np.abs(got - t).max(axis=1)); r_truth.append(np.abs(g.rows(i) - t).max(axis=1))
e_ref = np.concatenate(e_ref)
print("texts equal", texts == g.transcriptions, "argmax flips", flips)
bins = [0, 1e-4, 2e-4, 5e-4, 1e-3, 2e-3, 1]
print("rows: HIP vs reference   max %.3e  hist %s" % (e_ref.max(), np.histogram(e_ref, bins)[0].tolist()))
if e_truth:
    e_truth, r_truth = np.concatenate(e_truth), np.concatenate(r_truth)
    print("rows: HIP vs float64     max %.3e  rms %.3e  hist %s" % (e_truth.max(), np.sqrt(np.mean(e_truth ** 2)), np.histogram(e_truth, bins)[0].tolist()))
    print("rows: reference vs f64   max %.3e  rms %.3e  hist %s" % (r_truth.max(), np.sqrt(np.mean(r_truth ** 2)), np.histogram(r_truth, bins)[0].tolist()))
    k = np.argsort(-e_truth)[:6]
    print("worst rows (HIP vs f64, ref vs f64):", [(float("%.2e" % e_truth[j]), float("%.2e" % r_truth[j])) for j in k])
