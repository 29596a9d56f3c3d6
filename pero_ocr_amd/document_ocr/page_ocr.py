"""Caller-side adapter of the hot path: counterpart of the reference's PageOCR
(pero_ocr/document_ocr/page_parser.py:406-434).  It gathers the crops of a page's
text lines, calls `engine.process_lines` once for the whole page and writes the
four result fields back on each line object (transcription, logits, characters,
logit_coords - the TextLine contract of pero_ocr/core/layout.py:41-72).

The page/line classes are duck-typed: anything with `lines_iterator()` yielding
objects that carry `.crop` and `.id` works, so an unmodified pero-ocr PageLayout
can be passed in (INTEGRATION.md shows the one-line swap inside pero-ocr itself).
"""
from __future__ import annotations

from .. ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR


class LineCropper:
    """Counterpart of the reference's LineCropper (pero_ocr/document_ocr/page_parser.py:376-393): fills `line.crop`
    for every text line of a page - here with ONE GPU call for all lines (pero_ocr_amd/core/crop_engine.py)."""

    def __init__(self, config, config_path="", device_id: int = 0):
        """Build-specific key RESIDENT_CROPS (default no): `line.crop` becomes an array-like that stays in HBM
        (`_native.LazyCrop`) - PageOCR on the same GPU then recognises the lines without the crops crossing PCIe;
        anything that reads the pixels (np.asarray, indexing) gets a numpy copy on demand."""
        from ..core.crop_engine import EngineLineCropper
        self.crop_engine = EngineLineCropper(line_height=int(config["LINE_HEIGHT"]), poly=int(config["INTERP"]),
                                             scale=float(config["LINE_SCALE"]), device_id=device_id)
        self.resident = str(config.get("RESIDENT_CROPS", "no")).lower() in ("1", "yes", "true", "on")

    def process_page(self, img, page_layout):
        lines = list(page_layout.lines_iterator())
        crops = self.crop_engine.crop_lines(img, [(line.baseline, line.heights) for line in lines], resident=self.resident)
        for line, crop in zip(lines, crops):
            line.crop = crop
        return page_layout


class PageOCR:
    def __init__(self, config, device, config_path=""):
        """config: mapping with OCR_JSON (and optional USE_CPU / METHOD) like the [OCR] INI section."""
        import os
        json_file = config["OCR_JSON"]
        if not os.path.isabs(json_file):
            json_file = os.path.join(config_path, json_file)
        use_cpu = str(config.get("USE_CPU", "no")).lower() in ("1", "yes", "true", "on")
        if use_cpu:
            raise RuntimeError("USE_CPU is set: pero_ocr_amd has no CPU path")
        self.device = device
        if config.get("METHOD", "") == "pytorch_ocr-transformer":          # page_parser.py:413-414
            from ..ocr_engine.transformer_ocr_engine import TransformerEngineLineOCR
            self.ocr_engine = TransformerEngineLineOCR(json_file, self.device)
        else:
            self.ocr_engine = PytorchEngineLineOCR(json_file, self.device)

    def process_page(self, img, page_layout):
        self.process_pages([page_layout])
        return page_layout

    def process_pages(self, page_layouts, sharded=None):
        """The lines of SEVERAL pages through one `process_lines` call.  The recurrent layers advance one frame per
        dependent kernel whatever the number of lines (a page of 47 long lines keeps the GPU ~20 % busy there), and lines
        are recognised independently given their chunk's padded width - which is a function of the sorted widths, so
        the chunks of a page stream differ from the per-page ones exactly as they do when the reference's
        `process_lines` is handed more lines.  Results are written to the lines as process_page does.
        `sharded`: a sharding.ShardedLineOCR built over this engine - the stream's chunks are dealt to the ranks."""
        return self.process_pages_end(self.process_pages_begin(page_layouts, sharded))

    def process_pages_begin(self, page_layouts, sharded=None):
        """First half of process_pages: the lines' launches are enqueued (`process_lines_begin`), nothing is waited for.
        A page stream begins batch k + 1 before it ends batch k, so the recogniser's pipeline is not drained between
        batches; tickets are ended in the order they were begun."""
        lines = [line for layout in page_layouts for line in layout.lines_iterator()]
        for line in lines:
            if line.crop is None:
                raise Exception(f"Missing crop in line {line.id}.")
        # sharded (sharding.ShardedLineOCR over this engine, one process per GPU): every rank gets every transcription,
        # logits / logit_coords for the lines of its own chunks and None for the others (they stay on the producing rank)
        crops = [line.crop for line in lines]
        begin = getattr(self.ocr_engine, "process_lines_begin", None) if sharded is None else None
        if begin is None:                # sharded calls (one collective each) and engines without the two halves: in one piece
            recogniser = sharded if sharded is not None else self.ocr_engine
            return (page_layouts, lines, None, recogniser.process_lines(crops))
        return (page_layouts, lines, begin(crops), None)

    def process_pages_end(self, ticket):
        page_layouts, lines, job, result = ticket
        texts, logits, coords = result if job is None else self.ocr_engine.process_lines_end(job)
        for line, text, line_logits, line_coords in zip(lines, texts, logits, coords):
            line.transcription = text
            line.logits = line_logits
            line.characters = self.ocr_engine.characters
            line.logit_coords = line_coords
        # extra: confidences the GPU computed from the same sparse logits; PageParser.update_confidences
        # (page_parser.py:505-508) recomputes the same numbers on the host if it is left in place
        for line, conf in zip(lines, getattr(self.ocr_engine, "line_confidences", None) or []):
            if conf is not None:
                line.transcription_confidence = conf
        return page_layouts

    @property
    def provides_ctc_logits(self):
        return isinstance(self.ocr_engine, PytorchEngineLineOCR)
