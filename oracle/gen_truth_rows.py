"""ORACLE TOOLING — adds float64 "truth" rows to a golden fixture (default: c3).

The reference computes in float32; on long lines its own rounding noise reaches ~1e-3 on a few logits (measured: line
1530 of c3, class 63: reference vs float64 arithmetic 1.1e-3).  A 1e-3 logit bar against the reference alone would then
test the reference's noise, not this build.  This script runs the restated network (oracle/model_oracle.py, pinned
bit-exactly against the reference run by gen_golden.py) in FLOAT64 over the fixture's page stream, chunk by chunk as the
reference batches it, and stores the sampled rows as `rows64_delta16` = float16(reference row - float64 row): the
difference is < 2e-3, so float16 keeps it to < 1e-6 and the fixture stays small (conftest.Golden.rows64 rebuilds the
truth rows).  Tests then require
    |hip - truth| < 1e-3                      (this build against exact arithmetic)
    |hip - reference| < 1e-3 + |reference - truth|   (the reference's own deviation is not charged to the build)
Usage: python oracle/gen_truth_rows.py [fixture ...]      (c2 c2u c3 c4)
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN_DIR, Golden  # noqa: E402
from oracle import engine_oracle, model_oracle  # noqa: E402


def main(name):
    g = Golden(name)
    spec, weights, crops = g.spec(), g.weights(), g.crops()
    net = model_oracle.OracleNet(spec, weights).double()
    torch.set_num_threads(os.cpu_count() or 1)
    out = [None] * g.n
    st_colmax, st_colmean, st_rowlse = [None] * g.n, [None] * g.n, [None] * g.n
    for k, (ids, mw) in enumerate(g.plan):
        batch = engine_oracle.assemble_batch(crops, ids, spec.height, mw, 480 * g.batch_size)
        with torch.no_grad():
            x = (torch.from_numpy(np.ascontiguousarray(batch)).double() / 255.0).permute(0, 3, 1, 2)
            nct = net(x).numpy()
        for j, i in enumerate(ids):
            tc = nct[j].T                                   # [T, C] float64
            out[i] = tc[g.sample_rows[i]]
            # full-tensor statistics in exact arithmetic (1-Lipschitz in the max norm: tests/test_gpu_parity.py
            # _check_full_tensor_stats): per class max / mean over the frames, per frame logsumexp
            st_colmax[i], st_colmean[i] = tc.max(axis=0), tc.mean(axis=0)
            st_rowlse[i] = np.logaddexp.reduce(tc, axis=1)
        if k % 20 == 0:
            print(f"chunk {k}/{len(g.plan)}", flush=True)
    rows64 = np.concatenate(out)
    ref = np.concatenate([g.rows(i) for i in range(g.n)])
    print("reference vs float64 on the sampled rows: max %.3e, rows above 5e-4: %d of %d" %
          (np.abs(ref - rows64).max(), int((np.abs(ref - rows64).max(axis=1) > 5e-4).sum()), ref.shape[0]))
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    arrays = dict(np.load(path))
    arrays.pop("rows64_all", None)
    arrays["rows64_delta16"] = (ref.astype(np.float64) - rows64.astype(np.float64)).astype(np.float16)
    if "colmax" in arrays:      # the float64 statistics as float16(reference statistic - float64 statistic), like the rows
        arrays["colmax64_delta16"] = (arrays["colmax"].astype(np.float64) - np.stack(st_colmax)).astype(np.float16)
        arrays["colmean64_delta16"] = (arrays["colmean"].astype(np.float64) - np.stack(st_colmean)).astype(np.float16)
        arrays["rowlse64_delta16"] = (arrays["rowlse"].astype(np.float64) - np.concatenate(st_rowlse)).astype(np.float16)
        print("reference vs float64 statistics: colmax %.3e colmean %.3e rowlse %.3e" % tuple(
            float(np.abs(arrays[k].astype(np.float64)).max()) for k in ("colmax64_delta16", "colmean64_delta16", "rowlse64_delta16")))
    np.savez_compressed(path, **arrays)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["c3"]):
        main(nm)
