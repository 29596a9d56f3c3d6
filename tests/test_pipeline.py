"""Page image -> line crops -> transcriptions on the GPU: the rows of SURVEY.md section 8 chained the way the
reference's PageParser.process_page chains them (page_parser.py:515-531: line cropper, then OCR; the confidences of
update_confidences come with the OCR results).  Layout detection (row f-2) is not part of this build: the test
supplies the baselines."""
import json
import os

import numpy as np
import pytest

from oracle import crop_oracle
from pero_ocr_amd import synth

pytestmark = pytest.mark.gpu


class Dev:
    type, index = "cuda", 0


class Line:
    def __init__(self, i, baseline, heights):
        self.id, self.baseline, self.heights = f"r0-l{i}", np.array(baseline), heights
        self.crop = self.transcription = self.logits = self.characters = self.logit_coords = None
        self.transcription_confidence = None


class Layout:
    def __init__(self, lines):
        self.lines = lines

    def lines_iterator(self):
        return iter(self.lines)


def test_page_to_text(tmp_path, golden):
    from pero_ocr_amd.document_ocr.page_ocr import LineCropper, PageOCR
    g = golden("c1")
    rng = np.random.RandomState(8)
    page = np.full((1400, 1200, 3), 235, np.uint8)
    widths = [int(w) for w in rng.randint(200, 900, size=14)]
    glyphs = synth.make_crops(77, widths)
    lines = []
    for i, (w, img) in enumerate(zip(widths, glyphs)):
        y0, x0 = 40 + i * 95, int(rng.randint(20, 1180 - w))
        page[y0:y0 + 40, x0:x0 + w] = img
        tilt = int(rng.randint(-3, 4))
        lines.append(Line(i, [[x0, y0 + 30], [x0 + w // 2, y0 + 30 + tilt], [x0 + w, y0 + 30]], [30, 10]))
    layout = Layout(lines)
    LineCropper({"LINE_HEIGHT": str(g.height), "INTERP": "2", "LINE_SCALE": "1.0"}).process_page(page, layout)
    for ln in lines:
        want = crop_oracle.crop(page, ln.baseline, ln.heights, g.height, 1.0, 2)
        assert ln.crop.shape[0] == g.height and np.array_equal(ln.crop, want)
    ocr = PageOCR({"OCR_JSON": g.write_engine_json(tmp_path)}, Dev())
    ocr.process_page(page, layout)
    direct = ocr.ocr_engine.process_lines([ln.crop for ln in lines])[0]
    assert [ln.transcription for ln in lines] == direct
    assert all(ln.logits.shape[1] == len(g.characters) and ln.logit_coords[0] == 8 for ln in lines)
    assert all(0 < ln.transcription_confidence <= 1 for ln in lines)


def test_page_stream_with_layout_network_front(tmp_path, golden):
    """PageStream end to end on the GPU: layout network (its maps are computed, the baselines come from the generator - the
    post-processing is out of scope) + resident cropper on the helper thread, the recogniser on the lines of two pages per call.
    Every page must carry what `process_lines` returns for the batch it was part of (= the oracle engine on those lines), in order."""
    from oracle import engine_oracle, model_oracle
    from pero_ocr_amd import parsenet_spec
    from pero_ocr_amd.document_ocr.page_ocr import LineCropper, PageOCR
    from pero_ocr_amd.document_ocr.page_stream import PageStream
    from pero_ocr_amd.layout_engines import torch_parsenet
    g = golden("c1")
    pn_path = os.path.join(str(tmp_path), "pn.pocrp")
    torch_parsenet.save_blob(pn_path, parsenet_spec.generate_weights(5))
    parsenet = torch_parsenet.TorchParseNet(pn_path, Dev(), downsample=2, adaptive_downsample=False)
    cropper = LineCropper({"LINE_HEIGHT": str(g.height), "INTERP": "0", "LINE_SCALE": "1.0"})
    ocr = PageOCR({"OCR_JSON": g.write_engine_json(tmp_path)}, Dev())
    rng = np.random.RandomState(18)
    pages, boxes = [], {}
    for k in range(5):
        page = np.full((640, 1024, 3), 230, np.uint8)
        widths = [int(w) for w in rng.randint(150, 800, size=5)]
        rows = []
        for i, (w, img) in enumerate(zip(widths, synth.make_crops(200 + k, widths))):
            y0, x0 = 30 + i * 110, int(rng.randint(10, 1010 - w))
            page[y0:y0 + 40, x0:x0 + w] = img
            rows.append((x0, y0, w))
        pages.append(page)
        boxes[id(page)] = rows
    seen_maps = []

    def front(img):
        maps, ds = parsenet.get_maps_with_optimal_resolution(img)
        seen_maps.append(maps.shape)
        return Layout([Line(i, [[x0, y0 + 30], [x0 + w // 3, y0 + 31], [x0 + 2 * w // 3, y0 + 29], [x0 + w, y0 + 30]], [30, 10])
                       for i, (x0, y0, w) in enumerate(boxes[id(img)])])
    out = list(PageStream(front, cropper, ocr, pages_per_batch=2).process(iter(pages)))
    assert [id(img) for img, _ in out] == [id(p) for p in pages] and seen_maps == [(320, 512, 5)] * 5
    net = model_oracle.OracleNet(g.spec(), g.weights())
    for first in (0, 2, 4):                                      # batches of two pages (the last one alone)
        layouts = [lay for _img, lay in out[first:first + 2]]
        lines = [ln for lay in layouts for ln in lay.lines]
        crops = [ln.crop for ln in lines]
        # exactly what the engine returns for that batch on its own (the GPU path is deterministic) ...
        direct_t, direct_l, _ = ocr.ocr_engine.process_lines(crops, sparse_logits=False)
        assert [ln.transcription for ln in lines] == direct_t
        # ... and the oracle on those lines: logits within the tolerance, arg-max equal wherever the oracle's own top-2 margin
        # is not a near tie (these crops are not margin-selected like the fixtures', and the CPU oracle's last bits depend on
        # how oneDNN splits its work over the threads it gets)
        _t, want_l, _c, _ = engine_oracle.process_lines(lambda b: model_oracle.forward_logits(net, b), crops, ocr.ocr_engine.characters,
                                                        g.height, 480 * 8, sparse_logits=False)
        for got, want in zip(direct_l, want_l):
            got, want = np.asarray(got), np.asarray(want)
            assert float(np.max(np.abs(got - want))) < 1e-3
            srt = np.sort(want, axis=1)
            clear = (srt[:, -1] - srt[:, -2]) > 2e-3
            assert np.array_equal(np.argmax(got, axis=1)[clear], np.argmax(want, axis=1)[clear])
    for img, lay in out:
        for ln in lay.lines:
            assert np.array_equal(ln.crop, crop_oracle.crop(img, ln.baseline, ln.heights, g.height, 1.0, 0))


def test_page_stream_reproduces_the_reference_fixture(tmp_path, golden):
    """the page stream compared with the REFERENCE, not with the engine itself.  The 'pages' are groups of
    the ragged fixture's own crops (a cropper stand-in hands them out - resident in HBM for half of the pages, numpy for the
    others), so whatever PageStream batches together, every line must get the transcription the reference engine produced
    for it when the same lines went through its process_lines in one call - provided the stream hands the recogniser the same
    list of lines in one call too (pages_per_batch = all pages): chunking is a function of the whole list."""
    from pero_ocr_amd import _native
    from pero_ocr_amd.document_ocr.page_ocr import PageOCR
    from pero_ocr_amd.document_ocr.page_stream import PageStream
    g = golden("ragged")
    crops = g.crops()
    ocr = PageOCR({"OCR_JSON": g.write_engine_json(tmp_path)}, Dev())
    groups = [list(range(0, 5)), list(range(5, 6)), list(range(6, 12)), list(range(12, len(crops)))]
    pages = [np.zeros((8, 8, 3), np.uint8) + k for k in range(len(groups))]
    index = {id(p): k for k, p in enumerate(pages)}

    class FakeCropper:
        def process_page(self, img, layout):
            for ln, i in zip(layout.lines, groups[index[id(img)]]):
                ln.crop = crops[i]
            return layout

    def front(img):
        return Layout([Line(i, [[0, 0], [1, 0]], [1, 1]) for i in groups[index[id(img)]]])

    stream = PageStream(front, FakeCropper(), ocr, pages_per_batch=len(pages))
    seen = []
    for img, layout in stream.process(iter(pages)):
        for ln, i in zip(layout.lines, groups[index[id(img)]]):
            assert ln.transcription == g.transcriptions[i], f"line {i}: {ln.transcription!r} != reference {g.transcriptions[i]!r}"
            assert ln.logit_coords == g.logit_coords[i]
            seen.append(i)
    assert sorted(seen) == list(range(len(crops)))
