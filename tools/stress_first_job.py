"""Repro aid: fresh engines under load.  Three threads create an engine each, push two of the fixed jobs through it, compare with the
single-threaded expectation, destroy it, and start over, while a layout network runs on a fourth thread.
usage: python tools/stress_first_job.py [seconds=30] [threads=3]"""
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd import parsenet_spec, synth  # noqa: E402
from pero_ocr_amd.layout_engines import torch_parsenet  # noqa: E402
from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR  # noqa: E402


class Dev:
    type, index = "cuda", 0


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    chars = synth.make_charset(99)
    td = tempfile.mkdtemp()
    path = os.path.join(td, "ctc.json")
    json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent", "characters": chars, "net_name": "x",
               "net": {"weight_seed": 7}}, open(path, "w"))
    rng = np.random.RandomState(1)
    jobs = []
    for k in range(10):
        n = int(rng.choice([1, 5, 17, 48, 130, 300, 600]))
        widths = [int(w) for w in rng.choice([1, 33, 100, 257, 512, 640, 1000, 2100, 3900], size=n)]
        jobs.append(synth.make_crops(50 + k, widths))
    ref_engine = PytorchEngineLineOCR(path, Dev())
    expect = [ref_engine.process_lines(j, no_logits=True)[0] for j in jobs]
    del ref_engine
    stop = time.time() + seconds
    errors, counts = [], [0] * (n_threads + 1)

    def worker(t):
        r = np.random.RandomState(100 + t)
        try:
            while time.time() < stop and not errors:
                eng = PytorchEngineLineOCR(path, Dev())
                eng.pipeline_depth = 2 + t % 3
                for it in range(2):
                    k = int(r.randint(len(jobs)))
                    got = eng.process_lines(jobs[k], no_logits=bool(r.randint(2)))[0]
                    if got != expect[k]:
                        bad = [i for i, (a, b) in enumerate(zip(got, expect[k])) if a != b]
                        fb = eng.model.range_fallbacks() if hasattr(eng.model, "range_fallbacks") else None
                        errors.append(f"thread {t} engine #{counts[t]} call {it}: job {k} ({len(jobs[k])} lines) differs at lines {bad[:12]} of {len(bad)}; "
                                      f"widths {[jobs[k][i].shape[1] for i in bad[:12]]}; range fallbacks {fb}")
                        again = eng.process_lines(jobs[k], no_logits=True)[0]
                        errors.append(f"   the same job again on the same engine: {'equal to the expectation' if again == expect[k] else 'differs again'}")
                        return
                counts[t] += 1
                eng.model.close()
                del eng
        except Exception as exc:                            # noqa: BLE001
            errors.append(f"thread {t}: {type(exc).__name__}: {exc}")

    def front():
        try:
            pn = os.path.join(td, "pn.pocrp")
            torch_parsenet.save_blob(pn, parsenet_spec.generate_weights(3))
            net = torch_parsenet.TorchParseNet(pn, Dev(), downsample=4, adaptive_downsample=False)
            page = synth.make_page(5, 2048, 3072)
            while time.time() < stop and not errors:
                net.get_maps(page, 4)
                counts[n_threads] += 1
        except Exception as exc:                            # noqa: BLE001
            errors.append(f"front: {type(exc).__name__}: {exc}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)] + [threading.Thread(target=front)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    print(json.dumps({"engines_per_thread": counts[:n_threads], "layout_pages": counts[n_threads], "errors": errors}))
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
