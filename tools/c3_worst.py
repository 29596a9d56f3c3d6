#!/usr/bin/env python3
"""Where does the c3 fixture's worst |dlogit| come from?  Prints the lines with the largest sampled-row deviation."""
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
errs = []
for i in range(g.n):
    li = np.asarray(logits[i])
    d = np.abs(li[g.sample_rows[i]] - g.rows(i))
    errs.append((float(d.max()), i, g.widths[i], li.shape[0], int(np.argmax(d.max(axis=1))), int(np.argmax(d.max(axis=0))), float(np.abs(g.rows(i)).max())))
errs.sort(reverse=True)
print("texts equal", texts == g.transcriptions, " mean of per-line max", np.mean([e[0] for e in errs]))
for e in errs[:12]:
    print("err %.2e line %d width %d T %d row# %d class %d  max|logit| %.1f" % e)
hist = np.histogram([e[0] for e in errs], bins=[0, 1e-4, 2e-4, 5e-4, 1e-3, 2e-3, 1])[0]
print("hist of per-line max err [<1e-4,<2e-4,<5e-4,<1e-3,<2e-3,more]:", hist.tolist())
