"""Page stream: layout network -> layout post-processing -> line cropper -> line OCR over a sequence of pages, arranged
for one MI355X rather than page by page (the reference's PageParser.process_page, pero_ocr/document_ocr/page_parser.py:515-531,
runs the four stages of one page back to back on one device).

Two things a single page cannot give the GPU:
  * the front of page k+1 (layout network, cropper: their own HIP streams) runs on a helper thread while page k's lines
    are being recognised - the native calls release the GIL;
  * the recogniser gets the lines of `pages_per_batch` pages per `process_lines` call: its recurrent layers cost one
    dependent kernel per frame however many lines there are, so a lone page of long lines leaves most of the GPU idle.
Every page's results are the ones `process_lines` returns for the batch it was part of (see PageOCR.process_pages).
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable, Iterator, List, Tuple


class PageStream:
    def __init__(self, layout_front: Callable, cropper, page_ocr, pages_per_batch: int = 4, depth: int = 2):
        """layout_front(img) -> page layout (layout network + its post-processing); cropper.process_page(img, layout) fills
        the crops; page_ocr.process_pages(layouts) fills the text.  depth: batches the front may run ahead."""
        self.layout_front, self.cropper, self.page_ocr = layout_front, cropper, page_ocr
        self.pages_per_batch = max(1, int(pages_per_batch))
        self.depth = max(1, int(depth))

    def _front(self, pages: Iterable, out: "queue.Queue", stop: threading.Event):
        try:
            batch: List[Tuple[object, object]] = []
            for img in pages:
                if stop.is_set():
                    break
                layout = self.layout_front(img)
                self.cropper.process_page(img, layout)
                batch.append((img, layout))
                if len(batch) == self.pages_per_batch:
                    out.put(batch)
                    batch = []
            if batch:
                out.put(batch)
            out.put(None)
        except BaseException as exc:          # surfaces in the consumer
            out.put(exc)

    def process(self, pages: Iterable) -> Iterator[Tuple[object, object]]:
        """Yields (img, layout) in page order, every line carrying its transcription / logits / coords."""
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        worker = threading.Thread(target=self._front, args=(pages, q, stop), daemon=True)
        worker.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                self.page_ocr.process_pages([layout for _img, layout in item])
                yield from item
        finally:
            stop.set()
            while worker.is_alive():          # a consumer that stops early must not leave the producer blocked on put()
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            worker.join()
