import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, tempfile
from conftest import Golden
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR
class Dev: type, index = "cuda", 0
for name in ("c4", "c2"):
    g = Golden(name)
    td = tempfile.mkdtemp()
    eng = PytorchEngineLineOCR(g.write_engine_json(td), Dev(), batch_size=g.batch_size)
    texts, logits, coords = eng.process_lines(g.crops(), sparse_logits=False)
    flips = 0; worst = 0; minm = 1e9
    for i in range(g.n):
        li = np.asarray(logits[i]); am = np.argmax(li, 1); bad = am != g.argmax(i)
        flips += int(bad.sum())
        if bad.any(): minm = min(minm, float(g.margin(i)[bad].max()))
        worst = max(worst, float(np.max(np.abs(li[g.sample_rows[i]] - g.rows(i)))))
    print(name, "flips", flips, "max margin among flipped", minm, "worst sampled |dlogit|", worst, "texts equal", texts == g.transcriptions)
