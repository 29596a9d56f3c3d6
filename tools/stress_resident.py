"""Soak test of the resident recurrence under concurrency: several threads, each with its own engine (4 slots, pipeline depth
up to 4), push random launches of every size (1 .. 600 lines, 1 .. 3900 px) while a layout network and a cropper run on a
fourth thread - many resident recurrence kernels of different launches and engines are in flight together with convolutions.
Every call must return (no hand-off may time out) and equal the single-threaded result of the same lines.
usage: python tools/stress_resident.py [seconds=60] [threads=3]"""
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
    # (the engines warn about over-long lines with print(): three threads printing at once race inside CPython 3.10's TextIOWrapper -
    #  freed pending-bytes objects, i.e. arbitrary heap contents, end up in the output.  One lock around print keeps the log readable.)
    import builtins
    _print, _lock = builtins.print, threading.Lock()

    def locked_print(*a, **k):
        with _lock:
            _print(*a, **k)
    builtins.print = locked_print
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    chars = synth.make_charset(99)
    td = tempfile.mkdtemp()
    path = os.path.join(td, "ctc.json")
    json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent", "characters": chars, "net_name": "x",
               "net": {"weight_seed": 7}}, open(path, "w"))
    rng = np.random.RandomState(1)
    jobs = []
    for k in range(10):                                   # a fixed set of jobs with single-threaded expectations
        n = int(rng.choice([1, 5, 17, 48, 130, 300, 600]))
        widths = [int(w) for w in rng.choice([1, 33, 100, 257, 512, 640, 1000, 2100, 3900], size=n)]
        jobs.append(synth.make_crops(50 + k, widths))
    ref_engine = PytorchEngineLineOCR(path, Dev())
    expect = [ref_engine.process_lines(j, no_logits=True)[0] for j in jobs]
    del ref_engine
    stop = time.time() + seconds
    errors, counts = [], [0] * (n_threads + 1)

    def worker(t):
        try:
            eng = PytorchEngineLineOCR(path, Dev())
            eng.pipeline_depth = 2 + t % 3                 # 2 .. 4 launches in flight
            r = np.random.RandomState(100 + t)
            while time.time() < stop:
                k = int(r.randint(len(jobs)))
                got = eng.process_lines(jobs[k], no_logits=bool(r.randint(2)))[0]
                if got != expect[k]:
                    errors.append(f"thread {t}: job {k} differs")
                    return
                counts[t] += len(jobs[k])
        except Exception as exc:                            # noqa: BLE001
            errors.append(f"thread {t}: {type(exc).__name__}: {exc}")

    def front():
        try:
            pn = os.path.join(td, "pn.pocrp")
            torch_parsenet.save_blob(pn, parsenet_spec.generate_weights(3))
            net = torch_parsenet.TorchParseNet(pn, Dev(), downsample=4, adaptive_downsample=False)
            page = synth.make_page(5, 2048, 3072)
            while time.time() < stop:
                net.get_maps(page, 4)
                counts[n_threads] += 1
        except Exception as exc:                            # noqa: BLE001
            errors.append(f"front: {type(exc).__name__}: {exc}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)] + [threading.Thread(target=front)]
    t0 = time.time()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    print(json.dumps({"seconds": round(time.time() - t0, 1), "lines_per_thread": counts[:n_threads], "layout_pages": counts[n_threads],
                      "errors": errors}))
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
