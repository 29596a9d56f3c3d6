"""Soak of the page stream (two front workers, batches of pages in flight across process_lines calls): every page of a
long stream must come back with exactly the strings / coords its first pass produced.  usage: stress_page_stream.py [seconds] [fronts]"""
import contextlib, json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_ocr_amd import netspec, synth, parsenet_spec
from pero_ocr_amd.document_ocr.page_ocr import LineCropper, PageOCR
from pero_ocr_amd.document_ocr.page_stream import PageStream
from pero_ocr_amd.layout_engines import torch_parsenet

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n_fronts = int(sys.argv[2]) if len(sys.argv) > 2 else 2

class Dev:
    type, index = "cuda", 0

class Line:
    def __init__(self, i, baseline, heights):
        self.id, self.baseline, self.heights = f"l{i}", np.array(baseline), heights
        self.crop = self.transcription = self.logits = self.characters = self.logit_coords = None
        self.transcription_confidence = None

class Layout:
    def __init__(self, k, lines): self.k, self.lines = k, lines
    def lines_iterator(self): return iter(self.lines)

meta, spec, weights = bench.fixture_model("c2")
weights = dict(weights); weights["head.weight"] = weights["head.weight"] * np.float32(8); weights["head.bias"] = weights["head.bias"] * np.float32(8)
tmp = tempfile.mkdtemp()
netspec.save_blob(os.path.join(tmp, "weights.pocrw"), spec, weights)
json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "weights.pocrw", "characters": meta["characters"][:-1], "net_name": "b"}, open(os.path.join(tmp, "ocr.json"), "w"))
pn = os.path.join(tmp, "pn.pocrp"); torch_parsenet.save_blob(pn, parsenet_spec.generate_weights(20261001))
n_pages = 5
pages = [synth.make_page(900 + k, 3072, 4096) for k in range(n_pages)]
boxes = [synth.page_line_boxes(900 + k, 3072, 4096) for k in range(n_pages)]
index = {id(p): k for k, p in enumerate(pages)}

def make_front():
    net = torch_parsenet.TorchParseNet(pn, Dev(), downsample=4, adaptive_downsample=False)
    crop = LineCropper({"LINE_HEIGHT": "40", "INTERP": "2", "LINE_SCALE": "1.0", "RESIDENT_CROPS": "yes"})
    def front(img):
        k = index[id(img)]
        maps, _ds = net.get_maps_with_optimal_resolution(img)
        return Layout(k, [Line(i, [[x0, y0 + 30], [x0 + wd // 2, y0 + 30], [x0 + wd, y0 + 30]], [30, 10]) for i, (x0, y0, wd) in enumerate(boxes[k])]), float(maps[::64, ::64].sum())
    return (lambda img: front(img)[0]), crop

ocr = PageOCR({"OCR_JSON": os.path.join(tmp, "ocr.json")}, Dev())
fronts = [make_front() for _ in range(n_fronts)]
stream = PageStream(fronts[0][0], fronts[0][1], ocr, pages_per_batch=4, extra_fronts=fronts[1:])
first, n_done, t_end = {}, 0, time.time() + seconds

def endless():
    k = 0
    while time.time() < t_end:
        for _ in range(4):                # whole batches only: a partial last batch has another chunk plan (legitimately other last bits)
            yield pages[k % n_pages]      # 5 distinct pages in batches of 4: every page meets every position of a batch
            k += 1

with contextlib.redirect_stdout(sys.stderr):
    t0 = time.time()
    for img, lay in stream.process(endless()):
        got = ([l.transcription for l in lay.lines], [l.logit_coords for l in lay.lines], [int(l.logits.nnz) for l in lay.lines])
        key = (lay.k, n_done % 20)        # the batch composition (hence the chunk plan) repeats every 20 pages
        if key in first:
            assert first[key][0] == got[0], f"page {n_done}: transcriptions changed"
            assert first[key][1] == got[1] and first[key][2] == got[2], f"page {n_done}: coords / logits changed"
        else:
            first[key] = got
        n_done += 1
    dt = time.time() - t0
print(json.dumps({"pages": n_done, "seconds": round(dt, 1), "pages_per_s": round(n_done / dt, 2), "fronts": n_fronts, "distinct_keys": len(first),
                  "consumer_ms_per_page": {k: round(1e3 * v / max(1, n_done), 3) for k, v in stream.stats.items() if k != "batches"}, "result": "every repeat identical"}))
