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
