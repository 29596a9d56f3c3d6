"""Host side of the line recogniser: engine-JSON parsing, the reference's
width-sorted chunking, scatter of results back to input order, logit_coords and
logit sparsification.

Mirrors the public surface of the reference's BaseEngineLineOCR
(pero_ocr/ocr_engine/line_ocr_engine.py:16-177): same constructor arguments,
same attributes (`characters`, `batch_size`, `max_input_horizontal_pixels`,
`line_px_height`, `line_padding_px`, `embed_id`, `embed_num`, `config`, ...),
same `process_lines` signature, defaults and return contract.  `process_lines`
here is the CTC form; the "transformer" branches (:84-85,95-119,131-142,161-162) and
merge_transcriptions_and_logits / find_best_overlap (:180-211) are used by
transformer_ocr_engine.TransformerEngineLineOCR, which overrides process_lines.

Chunking is part of the numerical contract: a line's logits
depend on the padded width of its chunk, so `plan_chunks` reproduces
line_ocr_engine.py:79-90 exactly - descending stable width sort, chunk size
max(1, max_input_horizontal_pixels // ceil32(widest remaining)).
"""
from __future__ import annotations

import json
import os
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import sparse

from .softmax import softmax

# The reference prints its warnings (line_ocr_engine.py:125-127); here engines run on several threads (page stream: front workers
# next to the recogniser's thread, sharded ranks' helpers), and concurrent print() calls race inside CPython 3.10's TextIOWrapper -
# freed pending-bytes objects, i.e. arbitrary heap contents, reach the output (seen in tools/stress_resident.py: binary junk between
# the warnings of three threads, none with one lock around print).
_PRINT_LOCK = threading.Lock()


def print_warning(text: str) -> None:
    with _PRINT_LOCK:
        print(text)

SPARSE_PROB_THRESHOLD = 0.0001     # line_ocr_engine.py:170


@dataclass
class Chunk:
    """One device batch: which input lines, and the geometry they are padded to."""
    line_ids: List[int]
    max_width: int          # ceil32 of the widest line in the chunk
    w_pad: int              # columns actually fed to the network

    @property
    def frames(self) -> int:
        return (self.w_pad // 2) // 2


def ceil32(v: int) -> int:
    return -(-int(v) // 32) * 32


class ChunkPlan(Sequence):
    """The chunks of one process_lines call as a read-only sequence of `Chunk`.  The plan is held as arrays (first sorted
    position, size, ceil32 width and padded width of every chunk, over the width-sorted line order); a `Chunk` object is made
    when it is first asked for.  A rank of a sharded pass plans ALL chunks of the page stream but runs one eighth of them:
    it reads costs and payload geometry from the arrays (`sizes`, `w_pads`) and materialises only its own."""
    __slots__ = ("order", "starts", "sizes", "max_widths", "w_pads", "_made")

    def __init__(self, order: List[int], starts: np.ndarray, sizes: np.ndarray, max_widths: np.ndarray, w_pads: np.ndarray):
        self.order, self.starts, self.sizes, self.max_widths, self.w_pads = order, starts, sizes, max_widths, w_pads
        self._made: List[Optional[Chunk]] = [None] * len(starts)

    def __len__(self) -> int:
        return len(self._made)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self._made)))]
        c = self._made[k]
        if c is None:
            a = int(self.starts[k])
            c = self._made[k] = Chunk(self.order[a:a + int(self.sizes[k])], int(self.max_widths[k]), int(self.w_pads[k]))
        return c

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return f"ChunkPlan({list(self)!r})"


def plan_chunks(widths: Sequence[int], max_input_horizontal_pixels: int, line_padding_px: int = 32) -> ChunkPlan:
    """line_ocr_engine.py:79-90, 121-127 as a plan: stable sort by descending width, then chunks of
    max(1, limit // ceil32(widest remaining)) lines.  The sort and the rounding are numpy; the chunk boundaries are
    found run by run of equal ceil32 width (a chunk's size depends only on the width of its first line, so inside a run
    the starts are an arithmetic progression - ~30 steps instead of one per chunk): 0.1 ms for the 2048 lines / 343
    chunks of the c3 stream."""
    n = len(widths)
    limit, pad2 = int(max_input_horizontal_pixels), 2 * int(line_padding_px)
    if n == 0:
        z = np.zeros(0, np.int64)
        return ChunkPlan([], z, z, z, z)
    w = np.asarray(widths, dtype=np.int64)
    top = int(w.max())
    # (widths fit 16 bits on any real page: a stable sort of uint16 keys is numpy's radix sort, 4x the speed of the int64 one)
    order = np.argsort((top - w).astype(np.uint16), kind="stable") if 0 <= int(w.min()) and top < 65536 else np.argsort(-w, kind="stable")
    cw = -(-w[order] // 32) * 32                          # ceil32 of the widths, widest first
    run_first = np.flatnonzero(np.concatenate(([True], cw[1:] != cw[:-1])))
    run_end = np.concatenate((run_first[1:], [n])).tolist()
    run_w = cw[run_first].tolist()
    first, step, count, width, pos = [], [], [], [], 0
    for v, b in zip(run_w, run_end):                     # chunks that START inside the run of width v, i.e. at pos .. b - 1
        if pos >= b:
            continue
        if v == 0:               # the reference divides by ceil32(0) here (line_ocr_engine.py:87)
            raise ZeroDivisionError("zero-width line crop")
        take = limit // v or 1
        k = -(-(b - pos) // take)
        first.append(pos); step.append(take); count.append(k); width.append(v)
        pos += k * take
    count = np.asarray(count, dtype=np.int64)
    takes = np.repeat(np.asarray(step, dtype=np.int64), count)
    max_widths = np.repeat(np.asarray(width, dtype=np.int64), count)
    within = np.arange(int(count.sum()), dtype=np.int64) - np.repeat(np.cumsum(count) - count, count)
    starts = np.repeat(np.asarray(first, dtype=np.int64), count) + within * takes
    sizes = np.minimum(takes, n - starts)
    return ChunkPlan(order.tolist(), starts, sizes, max_widths, np.minimum(max_widths + pad2, limit))     # :121 then crop :125-127


@dataclass
class Launch:
    """A device launch: one or more consecutive reference chunks executed together.  Every line keeps
    the padded width of ITS chunk (that width is numerics); which chunks share a launch is only an
    execution choice, so small chunks are merged until a launch carries enough work to fill the GPU."""
    chunks: List[Chunk]

    @property
    def line_ids(self) -> List[int]:
        return [i for c in self.chunks for i in c.line_ids]

    @property
    def w_pads(self) -> List[int]:
        return [c.w_pad for c in self.chunks for _ in c.line_ids]

    @property
    def work(self) -> int:
        return sum(len(c.line_ids) * c.w_pad for c in self.chunks)


CHAIN_BOUND_FRAMES = 512              # a call whose longest line has at least this many frames (2048 px) ...
CHAIN_BOUND_MIN_WORK = 24 * 1024      # ... and at least three times this many padded columns is cut into three launches (see _begin_chunks)
LAUNCH_WORK_TARGET = 256 * 384          # padded pixel columns per launch (measured on the c3 stream: 82-98 k is 2-3 % better than 147 k; c5 neutral)


def launch_target(engine=None) -> int:
    """Work per launch: the engine's `launch_work_target` attribute, else POCR_LAUNCH_TARGET, else LAUNCH_WORK_TARGET."""
    t = getattr(engine, "launch_work_target", None)
    if t is None:
        t = int(os.environ.get("POCR_LAUNCH_TARGET", LAUNCH_WORK_TARGET))
    return max(1, int(t))


PIPELINE_DEPTH = 3          # launches in flight (see pipeline_depth)


def pipeline_depth(engine=None) -> int:
    """Launches in flight: the engine's `pipeline_depth` attribute, else POCR_PIPELINE_DEPTH, else 3; at most the slots the
    native engine has.  Three since the end of round 4 (profiles/r04_launch_timeline.txt): with two in flight the launches of a
    ragged stream complete in pairs and the GPU runs out of queued work in between - 2048 lines of 128..1024 px without logits
    95.2-96.3 -> 90.7-90.9 ms, with sparse logits 98.7 -> 94.0, a flat head's 61 MB of triplets per launch 91.5 -> 81.5; uniform
    512-px lines are unchanged within the run-to-run spread (78.7-84.1 against 80.8-81.3 ms)."""
    d = getattr(engine, "pipeline_depth", None)
    if d is None:
        d = int(os.environ.get("POCR_PIPELINE_DEPTH", PIPELINE_DEPTH))
    slots = getattr(getattr(engine, "model", None), "num_slots", 2)
    return max(1, min(int(d), int(slots)))


def plan_launches(chunks: Sequence[Chunk], target: int = LAUNCH_WORK_TARGET) -> List[Launch]:
    """Merge consecutive chunks (plan order = descending width) into ceil(total / target) launches of about equal work: a job
    a little larger than `target` (one page of long lines) becomes two balanced launches whose recurrent and convolutional
    phases overlap, not a full one and a remainder.  Cuts follow the CUMULATIVE work: launch j ends at the chunk boundary
    nearest to (j + 1) * total / n_launches, so no shortfall piles up in the last launch (every launch stays within one
    chunk's work of the mean; the largest launch sets a slot's activation high-water mark and the pipeline's tail)."""
    works = [len(ch.line_ids) * ch.w_pad for ch in chunks]
    total = sum(works)
    n_launches = max(1, -(-total // max(1, target)))
    out: List[Launch] = []
    cur: List[Chunk] = []
    done = 0                                   # work of the chunks already placed (closed launches + cur)
    for ch, w in zip(chunks, works):
        boundary = (len(out) + 1) * total / n_launches
        # cut BEFORE this chunk if that leaves the running total closer to the boundary than cutting after it would
        if cur and len(out) < n_launches - 1 and abs(done - boundary) <= abs(done + w - boundary):
            out.append(Launch(cur))
            cur = []
        cur.append(ch)
        done += w
    if cur:
        out.append(Launch(cur))
    return out


class BaseEngineLineOCR:
    def __init__(self, json_def, device, batch_size=8, model_type="ctc"):
        with open(json_def, "r", encoding="utf8") as f:
            self.config = json.load(f)
        cfg = self.config
        self.line_px_height = cfg["line_px_height"]
        self.line_vertical_scale = cfg["line_vertical_scale"]
        ckpt = cfg["checkpoint"]
        self.checkpoint = ckpt if os.path.isabs(ckpt) else os.path.realpath(
            os.path.join(os.path.dirname(json_def), ckpt))
        self.characters = tuple(cfg["characters"])
        self.net_name = cfg["net_name"]
        self.embed_num = int(cfg["embed_num"]) if "embed_num" in cfg else None
        self.embed_id = None
        if "embed_id" in cfg:
            self.embed_id = "mean" if cfg["embed_id"] == "mean" else int(cfg["embed_id"])
        self.max_line_width = int(cfg["max_line_width"]) if "max_line_width" in cfg else 1e10
        if model_type not in ("ctc", "transformer"):
            raise ValueError(f"unknown model_type {model_type!r}")
        self.model_type = model_type
        self.device = device
        self.batch_size = batch_size
        self.line_padding_px = 32
        self.max_input_horizontal_pixels = 480 * batch_size      # live-writable: drives the chunk plan

    # -- to be provided by the engine subclass -------------------------------------------
    def run_ocr(self, batch_data):
        raise NotImplementedError

    def _recognise_chunk(self, lines, chunk: Chunk, want_logits: bool):
        """-> (transcriptions, logits [n,T,C] or None).  Default: assemble the padded batch on
        the host and go through run_ocr (the reference's seam)."""
        batch = np.zeros([len(chunk.line_ids), self.line_px_height, chunk.max_width + 2 * self.line_padding_px, 3],
                         dtype=np.uint8)
        for row, i in zip(batch, chunk.line_ids):
            row[:, self.line_padding_px:self.line_padding_px + lines[i].shape[1], :] = lines[i]
        return self.run_ocr(batch[:, :, :chunk.w_pad])

    # -- public API -----------------------------------------------------------------------
    def process_lines(self, lines, sparse_logits=True, tight_crop_logits=False, no_logits=False):
        """Recognise a list of `uint8 [line_px_height, w_i, 3]` crops.

        Returns three lists in input order: transcriptions (str), logits
        (scipy.sparse.csc_matrix float32 [T_i, C]; dense ndarray if sparse_logits=False;
        None if no_logits) and logit_coords ([start, end] frame span of the un-padded
        line; [None, None] with tight_crop_logits; None if no_logits)."""
        return self.process_lines_end(self.process_lines_begin(lines, sparse_logits, tight_crop_logits, no_logits))

    def process_chunks(self, lines, chunks, sparse_logits=True, tight_crop_logits=False, no_logits=False):
        """The body of process_lines for a GIVEN subset of the reference's chunk plan of `lines` (all of it: process_lines;
        one rank's share: sharding.ShardedLineOCR - a line's result depends on its chunk's padded width, so a rank must run
        chunks of the plan over ALL lines, never a plan of its own lines).  Returns the three lists of process_lines, in
        input order, with None for the lines of chunks that were not given."""
        if getattr(self, "process_lines_end", None) is None or self.model_type != "ctc":
            raise TypeError("process_chunks is the CTC engine's call (chunks of the reference's CTC plan); the sequence-to-sequence "
                            "engine shards whole batches: sharding.ShardedSeq2SeqOCR / seq2seq_recogniser")
        return self.process_lines_end(self._begin_chunks(lines, chunks, sparse_logits, tight_crop_logits, no_logits))

    # -- the same call in two halves: a caller with a STREAM of process_lines calls (document_ocr.page_stream) begins call
    #    k + 1 before it ends call k, so the launches of consecutive calls share the engine's slots and the pipeline is not
    #    drained between calls.  What a call returns does not depend on what else is in flight (launches are independent).
    def process_lines_begin(self, lines, sparse_logits=True, tight_crop_logits=False, no_logits=False):
        """First half of process_lines: validates, plans and ENQUEUES (as far as the engine's slots allow); -> a ticket for
        process_lines_end.  Tickets must be ended in the order they were begun."""
        for i, line in enumerate(lines):
            if line.ndim != 3 or line.shape[0] != self.line_px_height or line.shape[2] != 3:
                raise ValueError(f"line {i}: expected a [{self.line_px_height}, w, 3] crop, got {line.shape}")
        chunks = plan_chunks([l.shape[1] for l in lines], self.max_input_horizontal_pixels, int(self.line_padding_px))
        return self._begin_chunks(lines, chunks, sparse_logits, tight_crop_logits, no_logits)

    def process_lines_end(self, job):
        """Second half: collects what is still in flight for this ticket; -> the three lists of process_lines."""
        if job.error is not None:
            raise job.error
        if job.done:
            raise RuntimeError("process_lines_end: this ticket was already ended")
        try:
            while job.open_launches:
                self._collect_oldest()
                if job.error is not None:
                    raise job.error
        except BaseException as exc:
            self._abort_inflight(exc)
            raise
        job.done = True
        # Side channel (not part of the reference's return contract): with GPU-built sparse logits the engine also
        # returns every line's transcription confidence, i.e. what PageParser.update_confidences would compute
        # from these logits (page_parser.py:485-496, 505-508).  None where it was not computed.
        self.line_confidences = job.confidences
        return job.transcriptions, job.logits_out, job.coords_out

    def _abort_inflight(self, exc):
        """A launch may still be in flight on any slot: fail every open ticket and leave the engine usable."""
        inflight = getattr(self, "_inflight", None)
        if inflight:
            for job, _launch, _handle, _sparse in inflight:
                if job.error is None:
                    job.error = exc if isinstance(exc, Exception) else RuntimeError(f"engine reset while this call was in flight: {exc!r}")
                job.open_launches = 0
            inflight.clear()
        reset = getattr(getattr(self, "model", None), "reset", None)
        if reset is not None:
            reset()

    def _collect_oldest(self):
        job, launch, handle, on_device = self._inflight.popleft()
        job.open_launches -= 1
        try:
            job.scatter(launch.line_ids, *self._collect_launch(handle), on_device=on_device)
        except BaseException as exc:
            job.error = exc if isinstance(exc, Exception) else RuntimeError(f"interrupted while this call was collected: {exc!r}")
            job.open_launches = 0
            raise

    def _begin_chunks(self, lines, chunks, sparse_logits, tight_crop_logits, no_logits):
        n = len(lines)
        sub = int(self.net_subsampling)
        pad = int(self.line_padding_px)
        device_sparse = sparse_logits and not no_logits and getattr(self, "supports_device_sparsify", False)
        job = _LinesJob(n)

        def scatter(line_ids, texts, chunk_logits, conf=None, on_device=False):
            """chunk_logits: per-line list (ragged launches, GPU-built csc or dense [T_i, C]) or [n, T, C] array."""
            transcriptions, logits_out, coords_out = job.transcriptions, job.logits_out, job.coords_out
            for k, i in enumerate(line_ids):
                transcriptions[i] = texts[k]
                if conf is not None:
                    job.confidences[i] = float(conf[k])
            if no_logits:
                return
            if on_device:               # chunk_logits is already a list of csc_matrix (built on the GPU)
                for k, i in enumerate(line_ids):
                    w = lines[i].shape[1]
                    coords_out[i] = [None, None] if tight_crop_logits else [pad // sub, (pad + w) // sub]
                    logits_out[i] = chunk_logits[k]
                return
            for k, i in enumerate(line_ids):
                w = lines[i].shape[1]
                first, last = pad // sub, (pad + w) // sub
                ll = chunk_logits[k]
                if tight_crop_logits:
                    ll = ll[first:last]
                    coords_out[i] = [None, None]
                else:
                    coords_out[i] = [first, last]
                if sparse_logits:
                    ll = sparse.csc_matrix(np.where(softmax(ll, axis=1) < SPARSE_PROB_THRESHOLD, np.float32(0), ll))
                logits_out[i] = ll
        job.scatter = scatter

        for chunk in chunks:
            if chunk.max_width + 2 * pad > chunk.w_pad:
                print_warning(f"WARNING: Line too long for OCR engine. Cropping from {chunk.max_width + 2 * pad} px "
                              f"down to {chunk.w_pad}.")
        if not hasattr(self, "_submit_launch"):          # engines without the asynchronous ragged path: chunk after chunk
            for chunk in chunks:
                scatter(chunk.line_ids, *self._recognise_chunk(lines, chunk, want_logits=not no_logits))
            return job

        # Software pipeline over LAUNCHES (merged chunks): launch k+1 is enqueued on another engine slot before launch k is
        # collected, so its GPU work overlaps launch k's read-back and the host-side assembly.  (The reference runs chunk
        # after chunk, line_ocr_engine.py:80-129; lines are independent given their padded width, so the results are the same.)
        # `depth` launches in flight (one engine slot each, pipeline_depth): with long lines the recurrent layers of a launch are a
        # chain of ~2 T dependent steps that leaves the GPU mostly idle - several chains side by side fill it; with short lines one
        # chain already hides behind the next launch's convolutions, and the third launch keeps work queued while this thread assembles
        # results.
        # The queue of launches in flight belongs to the ENGINE: launch number s runs on slot s % depth, which is free once
        # launch s - depth has been collected - whichever call (ticket) that one belongs to.
        from collections import deque
        if getattr(self, "_inflight", None) is None:
            self._inflight = deque()
            self._launch_seq = 0
        if not self._inflight:
            self._launch_seq = 0                         # an isolated call numbers its launches from slot 0, whatever ran before
        depth = pipeline_depth(self)
        launches = plan_launches(chunks, launch_target(self))
        # A lone call of one or two launches of LONG lines is bound by its recurrence chains (2 x T dependent steps per launch), not by
        # its convolutions: three launches in flight start the longest chain after a third of the convolutions instead of half
        # (one 4k x 3k page of 47 lines: 14.8-15.6 -> 13.9-14.6 ms, profiles/r04_sparse_blocks.txt).  Streams of calls and calls of
        # short lines keep the plan above (smaller launches cost them conv efficiency: c5 stream 64 -> 55 pages/s at half the target).
        if (not self._inflight and len(launches) <= 2 and getattr(self, "pipeline_depth", None) is None and
                "POCR_PIPELINE_DEPTH" not in os.environ and getattr(getattr(self, "model", None), "num_slots", 2) >= 3):
            total = sum(l.work for l in launches)
            longest = max((ch.frames for ch in chunks), default=0)
            if longest >= CHAIN_BOUND_FRAMES and total >= 3 * CHAIN_BOUND_MIN_WORK:
                launches = plan_launches(chunks, -(-total // 3))
                depth = 3
        if self._inflight and depth != getattr(self, "_inflight_depth", depth):
            while self._inflight:                        # the depth was changed between calls: start from an empty pipeline
                self._collect_oldest()
        self._inflight_depth = depth
        max_sparse_frames = getattr(self, "device_sparsify_max_frames", 0)
        try:
            for launch in launches:
                rows = None
                frames = [(wp // 2) // 2 for wp in launch.w_pads]
                # an engine may bound the frames per line its GPU sparsification takes (`device_sparsify_max_frames`; the HIP engine: no bound
                # since round 4): launches with longer lines take the dense read-back + host softmax / CSC instead
                launch_sparse = device_sparse and max(frames, default=0) <= max_sparse_frames
                if launch_sparse:
                    rows = (None, None)
                    if tight_crop_logits:
                        ws = [lines[i].shape[1] for i in launch.line_ids]
                        rows = ([min(pad // sub, f) for f in frames], [min((pad + w) // sub, f) for w, f in zip(ws, frames)])
                while len(self._inflight) >= depth:      # slot s % depth is free again once launch s - depth has been collected
                    self._collect_oldest()
                handle = self._submit_launch(lines, launch, not no_logits, self._launch_seq % depth, rows)
                self._launch_seq += 1
                job.open_launches += 1
                self._inflight.append((job, launch, handle, launch_sparse))
        except BaseException as exc:
            self._abort_inflight(exc)
            raise
        return job


class _LinesJob:
    """Ticket of one process_lines call: its result lists and the number of its launches still in flight."""
    __slots__ = ("transcriptions", "logits_out", "coords_out", "confidences", "open_launches", "scatter", "error", "done")

    def __init__(self, n: int):
        self.transcriptions: List[Optional[str]] = [None] * n
        self.logits_out: List[object] = [None] * n
        self.coords_out: List[Optional[list]] = [None] * n
        self.confidences: List[Optional[float]] = [None] * n
        self.open_launches = 0
        self.scatter = None
        self.error = None
        self.done = False


# ---- helpers of the "transformer" branch (over-long lines are recognised in overlapping parts) -------------

def levenshtein_distance(source, target) -> int:
    """Unit-cost edit distance (the quantity pero_ocr/sequence_alignment.py:4-13 returns for the default
    costs).  Row-wise DP; the insertion chain of a row is resolved with a running minimum."""
    tgt = np.asarray(list(target), dtype=object)
    m = len(tgt)
    ramp = np.arange(m + 1)
    dist = ramp.copy()
    for ch in source:
        cand = np.empty(m + 1, dtype=np.int64)
        cand[0] = dist[0] + 1
        if m:
            cand[1:] = np.minimum(dist[1:] + 1, dist[:-1] + (tgt != ch))
        dist = np.minimum.accumulate(cand - ramp) + ramp          # dist[j] = min_k<=j cand[k] + (j - k)
    return int(dist[-1])


def find_best_overlap(text1, text2) -> int:
    """Length i (1..min(len)) of the suffix of text1 / prefix of text2 with the lowest character error
    rate; the first such i wins; 0 when no rate is below 1 (line_ocr_engine.py:196-211).  The search is
    O(n^3); it runs in the native library (pocr_best_overlap), find_best_overlap_py is the same in numpy."""
    from .. import _native
    return _native.best_overlap(text1, text2)


def find_best_overlap_py(text1, text2) -> int:
    best_cer, best = 1, 0
    for i in range(1, min(len(text1), len(text2)) + 1):
        cer = levenshtein_distance(list(text1[-i:]), list(text2[:i])) / i
        if cer < best_cer:
            best_cer, best = cer, i
    return best


def merge_transcriptions_and_logits(transcription_parts, logits_parts):
    """Joins the parts of one line (line_ocr_engine.py:180-193).  Every part's logits are first cut to
    the length of its transcription; at each seam half of the best overlap is dropped on either side.
    The reference writes the left cut as `[:-overlap // 2]`, i.e. [: (-overlap) // 2]: the ceiling half,
    and an EMPTY left side when no overlap was found (overlap 0) - kept as is."""
    def stack(a, b):
        if sparse.issparse(a) or sparse.issparse(b):          # parts already sparsified on the GPU (row-wise operation)
            return sparse.vstack([a, b], format="csc")
        return np.concatenate([a, b], axis=0)

    def head(m, k):
        return m if m.shape[0] == k else m[:k]

    text = transcription_parts[0]
    logits = head(logits_parts[0], len(text))
    for nxt, nxt_logits in zip(transcription_parts[1:], logits_parts[1:]):
        nxt_logits = head(nxt_logits, len(nxt))
        overlap = find_best_overlap(text, nxt)
        left_end = (-overlap) // 2
        text = text[:left_end] + nxt[overlap // 2:]
        logits = stack(logits[:left_end], nxt_logits[overlap // 2:])
    return text, logits
