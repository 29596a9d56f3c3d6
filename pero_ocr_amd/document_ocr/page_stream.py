"""Page stream: layout network -> layout post-processing -> line cropper -> line OCR over a sequence of pages, arranged
for one MI355X rather than page by page (the reference's PageParser.process_page, pero_ocr/document_ocr/page_parser.py:515-531,
runs the four stages of one page back to back on one device).

Two things a single page cannot give the GPU:
  * the front of page k+1 (layout network, cropper: their own HIP streams) runs on a helper thread while page k's lines
    are being recognised - the native calls release the GIL;
  * batch k+1's launches are enqueued (`PageOCR.process_pages_begin`) before batch k's are collected, so the recogniser's
    pipeline of launches is not drained at every batch boundary (POCR_STREAM_OVERLAP=0: batch after batch);
  * the recogniser gets the lines of `pages_per_batch` pages per `process_lines` call: its recurrent layers cost one
    dependent kernel per frame however many lines there are, so a lone page of long lines leaves most of the GPU idle.
Every page's results are the ones `process_lines` returns for the batch it was part of (see PageOCR.process_pages).
"""
from __future__ import annotations

import os
import queue
import threading
import time
from typing import Callable, Iterable, Iterator, List, Tuple


class PageStream:
    def __init__(self, layout_front: Callable, cropper, page_ocr, pages_per_batch: int = 4, depth: int = 2, extra_fronts=()):
        """layout_front(img) -> page layout (layout network + its post-processing); cropper.process_page(img, layout) fills
        the crops; page_ocr.process_pages(layouts) fills the text.  depth: batches the front may run ahead.
        extra_fronts: further (layout_front, cropper) pairs - each pair gets a worker thread of its own and takes every n-th
        page (a pair owns device buffers, so two pages cannot share one).  A page's front is a chain of host steps and short
        dependent launches that wait their turn behind the recogniser's convolutions of EARLIER pages: next to a busy
        recogniser it takes twice as long as alone, and one worker cannot keep up - two fronts in flight can."""
        self.fronts = [(layout_front, cropper)] + [tuple(p) for p in extra_fronts]
        self.layout_front, self.cropper, self.page_ocr = layout_front, cropper, page_ocr
        self.pages_per_batch = max(1, int(pages_per_batch))
        self.depth = max(1, int(depth))
        self.overlap_batches = os.environ.get("POCR_STREAM_OVERLAP", "1") != "0"
        # where the consumer's time went (seconds): waiting for the front's next batch / inside the recogniser's calls
        self.stats = {"wait_front_s": 0.0, "ocr_s": 0.0, "batches": 0}

    def _front(self, pages: Iterable, out: "queue.Queue", stop: threading.Event):
        """Producer: pages in, batches of (img, layout) out, in page order.  Page i goes to front pair i % n; at most one page
        per pair is in flight, so the producer runs n pages ahead of the batch it is assembling."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        workers = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"pocr-front{i}") for i in range(len(self.fronts))]

        def one(pair, img):
            layout_front, cropper = pair
            layout = layout_front(img)
            cropper.process_page(img, layout)
            return img, layout

        try:
            batch: List[Tuple[object, object]] = []
            flying = deque()
            it = iter(pages)
            n, k, exhausted = len(self.fronts), 0, False
            while True:
                while not exhausted and len(flying) < n and not stop.is_set():
                    try:
                        img = next(it)
                    except StopIteration:
                        exhausted = True
                        break
                    flying.append(workers[k % n].submit(one, self.fronts[k % n], img))
                    k += 1
                if not flying:
                    break
                batch.append(flying.popleft().result())
                if len(batch) == self.pages_per_batch:
                    out.put(batch)
                    batch = []
                if stop.is_set():
                    exhausted = True
            if batch and not stop.is_set():
                out.put(batch)
            out.put(None)
        except BaseException as exc:          # surfaces in the consumer
            out.put(exc)
        finally:
            for w in workers:
                w.shutdown(wait=True)

    def process(self, pages: Iterable) -> Iterator[Tuple[object, object]]:
        """Yields (img, layout) in page order, every line carrying its transcription / logits / coords."""
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        worker = threading.Thread(target=self._front, args=(pages, q, stop), daemon=True)
        worker.start()
        begin = getattr(self.page_ocr, "process_pages_begin", None) if self.overlap_batches else None
        ahead = None                       # (batch, ticket) whose launches are in flight
        try:
            stats, clock = self.stats, time.perf_counter
            while True:
                t0 = clock()
                item = q.get()
                t1 = clock()
                stats["wait_front_s"] += t1 - t0
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                stats["batches"] += 1
                if begin is None:
                    self.page_ocr.process_pages([layout for _img, layout in item])
                    stats["ocr_s"] += clock() - t1
                    yield from item
                    continue
                # batch k + 1 is enqueued before batch k is collected: the recogniser's pipeline of launches runs through
                ticket = begin([layout for _img, layout in item])
                prev, ahead = ahead, (item, ticket)
                if prev is not None:
                    self.page_ocr.process_pages_end(prev[1])
                stats["ocr_s"] += clock() - t1
                if prev is not None:
                    yield from prev[0]
            if ahead is not None:
                t1 = clock()
                self.page_ocr.process_pages_end(ahead[1])
                stats["ocr_s"] += clock() - t1
                done, ahead = ahead[0], None
                yield from done
        finally:
            if ahead is not None:          # the consumer stopped early (or a batch failed): nothing may stay in flight
                try:
                    self.page_ocr.process_pages_end(ahead[1])
                except BaseException:
                    pass
            stop.set()
            while worker.is_alive():          # a consumer that stops early must not leave the producer blocked on put()
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            worker.join()
