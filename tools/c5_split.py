"""c5 stream: which side bounds it?  front only (layout net + cropper), recogniser only (4 pages of crops per call), both."""
import contextlib, json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pero_ocr_amd import netspec, synth, parsenet_spec
from pero_ocr_amd.document_ocr.page_ocr import LineCropper, PageOCR
from pero_ocr_amd.layout_engines import torch_parsenet

class Dev:
    type, index = "cuda", 0

class Line:
    def __init__(self, i, baseline, heights):
        self.id, self.baseline, self.heights = f"l{i}", np.array(baseline), heights
        self.crop = self.transcription = self.logits = self.characters = self.logit_coords = None
        self.transcription_confidence = None

class Layout:
    def __init__(self, lines): self.lines = lines
    def lines_iterator(self): return iter(self.lines)

meta, spec, weights = bench.fixture_model("c2")
weights = dict(weights); weights["head.weight"] = weights["head.weight"] * np.float32(8); weights["head.bias"] = weights["head.bias"] * np.float32(8)
tmp = tempfile.mkdtemp()
netspec.save_blob(os.path.join(tmp, "weights.pocrw"), spec, weights)
json.dump({"line_px_height": spec.height, "line_vertical_scale": 1.0, "checkpoint": "weights.pocrw", "characters": meta["characters"][:-1], "net_name": "b"}, open(os.path.join(tmp, "ocr.json"), "w"))
pn = os.path.join(tmp, "pn.pocrp"); torch_parsenet.save_blob(pn, parsenet_spec.generate_weights(20261001))
parsenet = torch_parsenet.TorchParseNet(pn, Dev(), downsample=4, adaptive_downsample=False)
cropper = LineCropper({"LINE_HEIGHT": "40", "INTERP": "2", "LINE_SCALE": "1.0"})
ocr = PageOCR({"OCR_JSON": os.path.join(tmp, "ocr.json")}, Dev())
pages = [synth.make_page(900 + k, 3072, 4096) for k in range(4)]
boxes = [synth.page_line_boxes(900 + k, 3072, 4096) for k in range(4)]
def layout_of(k): return Layout([Line(i, [[x0, y0 + 30], [x0 + wd // 2, y0 + 30], [x0 + wd, y0 + 30]], [30, 10]) for i, (x0, y0, wd) in enumerate(boxes[k])])
def front(k):
    parsenet.get_maps_with_optimal_resolution(pages[k]); lay = layout_of(k); cropper.process_page(pages[k], lay); return lay
with contextlib.redirect_stdout(sys.stderr):
    lays = [front(k) for k in range(4)]
    ocr.process_pages(lays)
    t0 = time.perf_counter()
    for r in range(3):
        for k in range(4): front(k)
    t_front = (time.perf_counter() - t0) / 12
    t0 = time.perf_counter()
    for r in range(3): ocr.process_pages(lays)
    t_ocr = (time.perf_counter() - t0) / 12
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for r in range(2): ocr.process_pages(lays)
    pr.disable()
print(json.dumps({"front_ms_per_page": round(1e3 * t_front, 2), "ocr_ms_per_page_4_pages_per_call": round(1e3 * t_ocr, 2)}))
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
