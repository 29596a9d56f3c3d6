"""PageStream (pero_ocr_amd/document_ocr/page_stream.py): page order, batching, error propagation - host logic, no GPU."""
import pytest

from pero_ocr_amd.document_ocr.page_stream import PageStream


class _Layout:
    def __init__(self, k):
        self.k, self.cropped, self.text = k, False, None


class _Cropper:
    def process_page(self, img, layout):
        layout.cropped = True


class _Ocr:
    def __init__(self):
        self.calls = []

    def process_pages(self, layouts):
        assert all(l.cropped for l in layouts)
        self.calls.append([l.k for l in layouts])
        for l in layouts:
            l.text = f"page {l.k}"


def test_stream_keeps_page_order_and_batches():
    ocr = _Ocr()
    st = PageStream(lambda img: _Layout(img), _Cropper(), ocr, pages_per_batch=4)
    out = list(st.process(range(10)))
    assert [img for img, _ in out] == list(range(10))
    assert [l.text for _, l in out] == [f"page {k}" for k in range(10)]
    assert ocr.calls == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    assert list(st.process([])) == []


def test_stream_surfaces_front_errors_and_early_stop():
    def front(img):
        if img == 5:
            raise ValueError("bad page")
        return _Layout(img)
    st = PageStream(front, _Cropper(), _Ocr(), pages_per_batch=2)
    got = []
    with pytest.raises(ValueError, match="bad page"):
        for img, _l in st.process(range(8)):
            got.append(img)
    assert got == [0, 1, 2, 3]
    it = PageStream(lambda img: _Layout(img), _Cropper(), _Ocr(), pages_per_batch=1, depth=1).process(range(100))
    assert next(it)[0] == 0
    it.close()                      # the producer must not stay blocked on a full queue
