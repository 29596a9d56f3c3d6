"""CPU tests of the two halves of process_lines (process_lines_begin / process_lines_end) and of the page stream that
uses them: a stand-in recogniser with the engine's asynchronous seam (_submit_launch / _collect_launch) records every
slot it is given, so the tests see the pipeline's invariants without a GPU.  Reference contract: process_lines
(pero_ocr/ocr_engine/line_ocr_engine.py:57-177) and PageOCR.process_page (document_ocr/page_parser.py:418-430)."""
import json

import numpy as np
import pytest

from pero_ocr_amd.document_ocr.page_ocr import PageOCR
from pero_ocr_amd.document_ocr.page_stream import PageStream
from pero_ocr_amd.ocr_engine.line_ocr_engine import BaseEngineLineOCR


class StandInEngine(BaseEngineLineOCR):
    """Labels are a function of the crop bytes and the chunk's padded width (like the real network's)."""
    pipeline_depth = 3
    launch_work_target = 2000

    def __init__(self, tmp_path, fail_at=None):
        path = tmp_path / "eng.json"
        path.write_text(json.dumps({"line_px_height": 8, "line_vertical_scale": 1.0, "checkpoint": "none",
                                    "characters": list("abcdefgh"), "net_name": "stand-in"}))
        super().__init__(str(path), device=None, batch_size=1)
        self.net_subsampling = 4
        self.busy = {}                 # slot -> launch number in flight
        self.events = []
        self.submitted = 0
        self.fail_at = fail_at
        self.model = type("M", (), {"num_slots": 4, "reset": lambda _s: self.busy.clear()})()

    def _submit_launch(self, lines, launch, want_logits, slot, sparse_rows=None):
        assert slot not in self.busy, f"slot {slot} handed out while launch {self.busy.get(slot)} is still in flight"
        self.busy[slot] = self.submitted
        self.events.append(("submit", self.submitted, slot))
        texts = ["".join(self.characters[(int(lines[i].sum()) + wp + k) % 8] for k in range(1 + lines[i].shape[1] % 5))
                 for i, wp in zip(launch.line_ids, launch.w_pads)]
        logits = [np.full(((wp // 2) // 2, 9), float(lines[i].shape[1]), np.float32) for i, wp in zip(launch.line_ids, launch.w_pads)]
        self.submitted += 1
        return (self.submitted - 1, slot, texts, logits if want_logits else None)

    def _collect_launch(self, handle):
        seq, slot, texts, logits = handle
        assert self.busy.pop(slot) == seq
        self.events.append(("collect", seq, slot))
        if self.fail_at is not None and seq == self.fail_at:
            raise RuntimeError("device lost (simulated)")
        return texts, logits


def _lines(seed, n):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 255, (8, int(w), 3), dtype=np.uint8) for w in rng.integers(1, 400, n)]


def _same(a, b):
    assert a[0] == b[0] and a[2] == b[2]
    for x, y in zip(a[1], b[1]):
        assert (x is None and y is None) or np.array_equal(x, y)


def test_two_calls_in_flight_return_what_two_calls_in_sequence_return(tmp_path):
    la, lb = _lines(1, 23), _lines(2, 31)
    seq = StandInEngine(tmp_path)
    want_a = seq.process_lines(la, sparse_logits=False)
    want_b = seq.process_lines(lb, sparse_logits=False, tight_crop_logits=True)
    eng = StandInEngine(tmp_path)
    ta = eng.process_lines_begin(la, sparse_logits=False)
    in_flight_of_a = ta.open_launches
    tb = eng.process_lines_begin(lb, sparse_logits=False, tight_crop_logits=True)
    assert 0 < in_flight_of_a <= 3 and tb.open_launches > 0          # a's tail stayed in flight until b needed the slots
    _same(eng.process_lines_end(ta), want_a)
    _same(eng.process_lines_end(tb), want_b)
    assert not eng.busy and not eng._inflight
    # launches are collected in the order they were submitted, and at most `depth` are in flight
    open_now, last = 0, -1
    for kind, s, _slot in eng.events:
        if kind == "submit":
            open_now += 1
            assert open_now <= 3
        else:
            open_now -= 1
            assert s == last + 1
            last = s
    with pytest.raises(RuntimeError):
        eng.process_lines_end(ta)
    # a plain call while a ticket is open: both complete
    tc = eng.process_lines_begin(la, no_logits=True)
    _same(eng.process_lines(lb, sparse_logits=False, tight_crop_logits=True), want_b)
    t, l, c = eng.process_lines_end(tc)
    assert t == want_a[0] and l == [None] * len(la) and c == [None] * len(la)


def test_failing_launch_fails_every_open_ticket_and_leaves_the_engine_usable(tmp_path):
    la, lb = _lines(3, 40), _lines(4, 12)
    probe = StandInEngine(tmp_path)
    probe.process_lines(la, no_logits=True)
    n_a = probe.submitted                                    # launches of the first call
    assert n_a > 3
    eng = StandInEngine(tmp_path, fail_at=n_a - 1)           # its last launch fails: collected while the second call is begun
    ta = eng.process_lines_begin(la, sparse_logits=False)
    with pytest.raises(RuntimeError, match="device lost"):
        eng.process_lines_begin(lb, sparse_logits=False)
    with pytest.raises(RuntimeError, match="device lost"):
        eng.process_lines_end(ta)
    assert not eng._inflight and not eng.busy
    eng.fail_at = None
    _same(eng.process_lines(lb, sparse_logits=False), StandInEngine(tmp_path).process_lines(lb, sparse_logits=False))


class _Line:
    def __init__(self, i, crop):
        self.id, self.crop = f"l{i}", crop
        self.transcription = self.logits = self.characters = self.logit_coords = None


class _Layout:
    def __init__(self, lines): self.lines = lines
    def lines_iterator(self): return iter(self.lines)


class _Cropper:
    def process_page(self, img, layout): return layout


def test_page_stream_with_batches_in_flight_equals_batch_after_batch(tmp_path, monkeypatch):
    def run(overlap):
        monkeypatch.setenv("POCR_STREAM_OVERLAP", "1" if overlap else "0")
        ocr = PageOCR.__new__(PageOCR)
        ocr.ocr_engine = StandInEngine(tmp_path)
        pages = [_Layout([_Line(i, c) for i, c in enumerate(_lines(100 + k, 5 + k))]) for k in range(7)]
        stream = PageStream(lambda img: img, _Cropper(), ocr, pages_per_batch=2)
        out = [lay for _img, lay in stream.process(iter(pages))]
        assert out == pages and not ocr.ocr_engine.busy
        return [[(l.transcription, l.logit_coords, l.logits.nnz) for l in lay.lines] for lay in out], ocr.ocr_engine.events
    (a, ev_a), (b, ev_b) = run(True), run(False)
    assert a == b and all(t is not None for page in a for t, _c, _n in page)
    # batch after batch drains the pipeline at every batch boundary; the overlapped stream does not
    def drained(events):
        open_now, n = 0, 0
        for kind, _s, _slot in events:
            open_now += 1 if kind == "submit" else -1
            n += open_now == 0
        return n
    assert drained(ev_b) >= 4 and drained(ev_a) <= 2
    # a consumer that stops early leaves nothing in flight
    ocr = PageOCR.__new__(PageOCR)
    ocr.ocr_engine = StandInEngine(tmp_path)
    monkeypatch.setenv("POCR_STREAM_OVERLAP", "1")
    it = PageStream(lambda img: img, _Cropper(), ocr, pages_per_batch=1).process(
        iter([_Layout([_Line(0, c) for c in _lines(7, 9)]) for _ in range(5)]))
    next(it); it.close()
    assert not ocr.ocr_engine.busy and not ocr.ocr_engine._inflight


def test_two_front_workers_keep_page_order_and_their_own_buffers(tmp_path):
    """Page i goes to front pair i % 2; results come back in page order whatever the fronts' speed."""
    import threading, time
    seen = {0: [], 1: []}

    def front(which):
        def layout_front(img):
            time.sleep(0.002 * (1 + (img.k * 7) % 3))          # uneven fronts: the later page may finish first
            seen[which].append((img.k, threading.current_thread().name))
            return img
        return layout_front

    class _Crop:
        def __init__(self, which): self.which, self.pages = which, []
        def process_page(self, img, layout): self.pages.append(img.k); return layout

    ocr = PageOCR.__new__(PageOCR)
    ocr.ocr_engine = StandInEngine(tmp_path)
    pages = [_Layout([_Line(i, c) for i, c in enumerate(_lines(300 + k, 4))]) for k in range(9)]
    for k, pg in enumerate(pages):
        pg.k = k
    c0, c1 = _Crop(0), _Crop(1)
    stream = PageStream(front(0), c0, ocr, pages_per_batch=2, extra_fronts=[(front(1), c1)])
    out = [lay for _img, lay in stream.process(iter(pages))]
    assert out == pages and all(l.transcription is not None for pg in out for l in pg.lines)
    assert c0.pages == [0, 2, 4, 6, 8] and c1.pages == [1, 3, 5, 7]
    assert len({name for _k, name in seen[0]}) == 1 and len({name for _k, name in seen[1]}) == 1
    assert {name for _k, name in seen[0]} != {name for _k, name in seen[1]}
    # one front, same pages: same texts
    ocr2 = PageOCR.__new__(PageOCR)
    ocr2.ocr_engine = StandInEngine(tmp_path)
    pages2 = [_Layout([_Line(i, c) for i, c in enumerate(_lines(300 + k, 4))]) for k in range(9)]
    out2 = [lay for _img, lay in PageStream(lambda img: img, _Cropper(), ocr2, pages_per_batch=2).process(iter(pages2))]
    assert [[l.transcription for l in pg.lines] for pg in out] == [[l.transcription for l in pg.lines] for pg in out2]
