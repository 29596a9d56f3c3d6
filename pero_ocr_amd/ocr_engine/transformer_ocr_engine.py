"""MI355X-native sequence-to-sequence (transformer) line-recognition engine with the
reference's engine surface.

Drop-in for pero_ocr/ocr_engine/transformer_ocr_engine.py: class name, constructor
`(json_def, device, batch_size=4)`, attributes (`characters` + boundary + ignore symbols,
`sentence_boundary_ind`, `ignore_ind`), `run_ocr(batch_data) -> (strings, float32 [n, steps, C])`,
`decode(labels)`, and `process_lines` with the "transformer" behaviour of
BaseEngineLineOCR.process_lines (pero_ocr/ocr_engine/line_ocr_engine.py:57-177): width-sorted batches
with the max_line_width clamp (:84-85), over-long lines recognised in overlapping parts (:95-119) and
merged again (:131-142, :180-211), logit_coords = [0, len(transcription)] (:161-162).

Where the reference runs `net.encode` and a Python loop of cached decoder calls per batch
(transformer_ocr_engine.py:49-89), this engine hands the uint8 crops of SEVERAL batches to the HIP
library (include/pocr.h, pocr_s2s_*): every line keeps the padded width of its own batch (numerics),
batches of one launch are decoded side by side and each ends where the reference's loop would end for it.

Engine JSON: the reference's keys; `net_name` is the network JSON of transformer.build_net (:13-20:
dim_model, dim_ff, heads, encoder_layers, decoder_layers, conv_subsampling - only [8, 4] is built);
`checkpoint` names a POCRW001 weight blob (tools/export_weights.py converts a reference state_dict).
If that file does not exist and the build-specific key "net": {"weight_seed": ..} is present, seeded
synthetic weights are generated.  There is no CPU fallback.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import sparse

from .. import _native, netspec
from .line_ocr_engine import (BaseEngineLineOCR, SPARSE_PROB_THRESHOLD, ceil32, merge_transcriptions_and_logits, print_warning)
from .pytorch_ocr_engine import _device_index
from .softmax import softmax

S2S_DEPTH = 3                    # launches whose decoding loops run side by side (one worker thread and one engine slot each)
MIN_INPUT_WIDTH = 1088           # transformer_ocr_engine.py:36-40: narrower batches are centred in 1088 columns
# device lines decoded side by side in one launch: a decoding step is latency-bound, so wide launches amortise it (round 1: 2.3k / 2.7k /
# 2.9k lines/s for 64 / 128 / 256; round 4, one decoding loop at a time: 6.9k / 7.5k / 7.6k / 7.7k for 256 / 384 / 512 / 1024)
LAUNCH_MAX_LINES = int(os.environ.get("POCR_S2S_MAX_LINES", 512))
LAUNCH_MAX_COLUMNS = LAUNCH_MAX_LINES * 1088


class _Batch:
    """One reference batch: its input lines, their parts and the geometry the parts are padded to."""

    def __init__(self, line_ids: List[int], max_width: int):
        self.line_ids = line_ids
        self.max_width = max_width
        self.parts: List[Tuple[int, int, int]] = []     # (line id, first column, end column)
        self.spans: List[int] = []                      # parts per line
        self.w_batch = 0                                # columns of the reference's batch_data
        self.w_pad = 0                                  # columns the network sees (>= 1088)
        self.pad_left = 0


def split_spans(width: int, max_line_width) -> List[Tuple[int, int]]:
    """Column spans of the parts of one line (line_ocr_engine.py:96-113): parts of max_line_width
    columns, consecutive parts overlapping by a quarter of it."""
    if width <= max_line_width:
        return [(0, width)]
    step = max_line_width - max_line_width // 4
    spans, start, end = [], 0, max_line_width
    while end < width:
        spans.append((start, end))
        start, end = start + step, end + step
    spans.append((start, min(end, width)))
    return spans


def plan_batches(widths: Sequence[int], max_input_horizontal_pixels: int, max_line_width, line_padding_px: int = 32) -> List[_Batch]:
    order = sorted(range(len(widths)), key=lambda i: -int(widths[i]))
    out, pos = [], 0
    while pos < len(order):
        max_width = ceil32(widths[order[pos]])
        max_width = int(min(max_width, max_line_width + 2 * line_padding_px))        # :84-85
        if max_width == 0:
            raise ZeroDivisionError("zero-width line crop")
        take = max(1, int(max_input_horizontal_pixels) // max_width)
        b = _Batch(order[pos:pos + take], max_width)
        for i in b.line_ids:
            spans = split_spans(int(widths[i]), max_line_width)
            b.parts += [(i, a, e) for a, e in spans]
            b.spans.append(len(spans))
        b.w_batch = min(max_width + 2 * line_padding_px, int(max_input_horizontal_pixels))   # :121, :125-127
        b.w_pad = max(b.w_batch, MIN_INPUT_WIDTH)
        b.pad_left = line_padding_px + ((MIN_INPUT_WIDTH - b.w_batch) // 2 if b.w_batch < MIN_INPUT_WIDTH else 0)
        out.append(b)
        pos += take
    return out


def plan_launches(batches: Sequence[_Batch]) -> List[List[_Batch]]:
    """Consecutive batches grouped into device launches.  A launch decodes all its lines side by side and a decoding
    step costs about the same for 2 lines as for 256, so the greedy packing (fill up to the caps) is evened out
    afterwards: the same number of launches, but of similar size instead of full ones and a small remainder."""
    def pack(max_lines, max_cols):
        out, cur, lines, cols = [], [], 0, 0
        for b in batches:
            n, c = len(b.parts), len(b.parts) * b.w_pad
            if cur and (lines + n > max_lines or cols + c > max_cols):
                out.append(cur)
                cur, lines, cols = [], 0, 0
            cur.append(b)
            lines += n
            cols += c
        if cur:
            out.append(cur)
        return out

    greedy = pack(LAUNCH_MAX_LINES, LAUNCH_MAX_COLUMNS)
    k = len(greedy)
    if k <= 1:
        return greedy
    total_lines = sum(len(b.parts) for b in batches)
    total_cols = sum(len(b.parts) * b.w_pad for b in batches)
    biggest = max(len(b.parts) for b in batches)
    for slack in (1.0, 1.1, 1.25, 1.5):
        even = pack(min(LAUNCH_MAX_LINES, int(total_lines / k * slack) + biggest),
                    min(LAUNCH_MAX_COLUMNS, int(total_cols / k * slack) + biggest * max(b.w_pad for b in batches)))
        if len(even) == k:
            return even
    return greedy


class TransformerEngineLineOCR(BaseEngineLineOCR):
    # the two halves of the CTC form's process_lines (BaseEngineLineOCR.process_lines_begin / _end) do not apply here:
    # this engine overrides process_lines with the reference's "transformer" branches (callers look for None)
    process_lines_begin = None
    process_lines_end = None

    def __init__(self, json_def, device, batch_size=4):
        super().__init__(json_def, device, batch_size=batch_size, model_type="transformer")
        self.characters = list(self.characters) + ["\u200B", ""]            # transformer_ocr_engine.py:16
        self.sentence_boundary_ind = len(self.characters) - 2
        self.ignore_ind = len(self.characters) - 1
        net = json.loads(self.net_name) if isinstance(self.net_name, str) else dict(self.net_name)
        sub = [int(v) for v in net.get("conv_subsampling", [8, 4])]
        if sub != [netspec.NET_SUBSAMPLING_H, netspec.NET_SUBSAMPLING_W]:
            raise NotImplementedError(f"conv_subsampling {sub}: only [8, 4] is built for MI355X")
        build_cfg = dict(self.config.get("net", {}))
        if os.path.exists(self.checkpoint):
            spec, weights = netspec.load_blob(self.checkpoint)
        elif "weight_seed" in build_cfg:
            spec = netspec.NetSpec(num_classes=len(self.characters), height=int(self.line_px_height),
                                   conv_out=int(net["dim_model"]), arch=netspec.ARCH_S2S,
                                   sa_layers=int(net["encoder_layers"]), sa_heads=int(net["heads"]),
                                   sa_ff=int(net["dim_ff"]), dec_layers=int(net["decoder_layers"]))
            gen = {k: build_cfg[k] for k in ("boundary_bias", "embed_gain", "walk_gain", "walk_stride") if k in build_cfg}
            weights = netspec.generate_weights(spec, int(build_cfg["weight_seed"]), **gen)
        else:
            raise FileNotFoundError(f"weight blob {self.checkpoint} not found and no net.weight_seed in the engine JSON")
        if spec.arch != netspec.ARCH_S2S:
            raise ValueError(f"{self.checkpoint}: not a sequence-to-sequence model (arch {spec.arch})")
        if spec.num_classes != len(self.characters):
            raise ValueError(f"model has {spec.num_classes} classes, engine JSON implies {len(self.characters)} "
                             "(characters + boundary + ignore)")
        if spec.height != int(self.line_px_height):
            raise ValueError(f"model height {spec.height} != line_px_height {self.line_px_height}")
        self.net_spec = spec
        self.net = _native.NativeEngine(spec, netspec.pack_weights(spec, weights), _device_index(self.device))

    supports_device_sparsify = True

    # ---- label post-processing (transformer_ocr_engine.py:91-111) ---------------------------------------
    def postprocess_decoded(self, transcripts, ignore_ind, sentence_boundary_ind) -> List[np.ndarray]:
        out = []
        for row in transcripts:
            row = np.asarray(row)
            stop = np.flatnonzero(row == sentence_boundary_ind)
            if stop.size:
                row = row[:stop[0]]
            out.append(row[row != ignore_ind])
        return out

    def decode(self, labels) -> List[str]:
        return ["".join(self.characters[int(c)] for c in row) for row in labels]

    # ---- device calls ------------------------------------------------------------------------------------
    def _submit(self, slot: int, images: Sequence[np.ndarray], w_pads, pad_lefts, batch_first):
        flat = [np.ascontiguousarray(im, dtype=np.uint8).reshape(-1) for im in images]
        widths = np.array([im.shape[1] for im in images], dtype=np.int32)
        sizes = np.array([f.size for f in flat], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        pool = np.concatenate(flat) if flat else np.zeros(0, np.uint8)
        self.net.s2s_stage(slot, pool, offsets, widths, w_pads, pad_lefts)
        self.net.s2s_launch(slot, batch_first)

    def _finish(self, slot: int, batch_first, want_logits: bool, device_sparse: bool = False):
        """-> per batch: (label arrays per device line, logits).  logits: [lines, steps_b, C] array, None, or
        (device_sparse) a list of csc_matrix [len(transcription_i), C] built on the GPU (pocr_s2s_sparse)."""
        steps, tokens, logits = self.net.s2s_decode(slot, want_logits=want_logits and not device_sparse)
        out, all_labels = [], []
        for b in range(len(batch_first) - 1):
            lo, hi, sb = int(batch_first[b]), int(batch_first[b + 1]), int(steps[b])
            kept = tokens[lo:hi, :max(sb - 1, 0)]         # partial_transcripts[1:] (:82-84): the last sample is never appended
            labels = self.postprocess_decoded(kept, self.ignore_ind, self.sentence_boundary_ind)
            all_labels += labels
            out.append((labels, logits[lo:hi, :sb] if logits is not None else None))
        if device_sparse and want_logits:
            # rows kept per line = len(transcription) (merge_transcriptions_and_logits cuts the logits there, :183)
            rows = [len("".join(self.characters[int(c)] for c in lab)) for lab in all_labels]
            data, indices, indptr, line_off = self.net.s2s_sparse(slot, rows, SPARSE_PROB_THRESHOLD)
            C_ = indptr.shape[1] - 1
            from .pytorch_ocr_engine import _csc_from_device
            mats = [_csc_from_device(data[int(line_off[i]):int(line_off[i + 1])], indices[int(line_off[i]):int(line_off[i + 1])],
                                     indptr[i], (rows[i], C_)) for i in range(len(rows))]
            out = [(labels, mats[int(batch_first[b]):int(batch_first[b + 1])]) for b, (labels, _l) in enumerate(out)]
        return out

    def transcribe_batch(self, inputs, is_cached=True):
        """uint8 [n, 3, H, W] (the reference's layout at this seam, :49) -> (label arrays, logits [n, steps, C])"""
        batch = np.ascontiguousarray(np.transpose(np.asarray(inputs), (0, 2, 3, 1)), dtype=np.uint8)
        n, _h, w, _c = batch.shape
        self._submit(0, list(batch), [w] * n, [0] * n, [0, n])
        (labels, logits), = self._finish(0, [0, n], True)
        return labels, logits

    def run_ocr(self, batch_data) -> Tuple[List[str], np.ndarray]:
        """uint8 [n, H, W, 3] -> (strings, float32 [n, steps, C]); batches narrower than 1088 px are centred
        in 1088 zero columns first (transformer_ocr_engine.py:32-47)."""
        b = np.asarray(batch_data)
        n, _h, w, _c = b.shape
        w_pad = max(w, MIN_INPUT_WIDTH)
        left = (MIN_INPUT_WIDTH - w) // 2 if w < MIN_INPUT_WIDTH else 0
        self._submit(0, list(b), [w_pad] * n, [left] * n, [0, n])
        (labels, logits), = self._finish(0, [0, n], True)
        return self.decode(labels), logits

    # ---- public API --------------------------------------------------------------------------------------
    def process_lines(self, lines, sparse_logits=True, tight_crop_logits=False, no_logits=False):
        n = len(lines)
        for i, line in enumerate(lines):
            if line.ndim != 3 or line.shape[0] != self.line_px_height or line.shape[2] != 3:
                raise ValueError(f"line {i}: expected a [{self.line_px_height}, w, 3] crop, got {line.shape}")
        if tight_crop_logits and not no_logits:
            # the reference slices with self.net_subsampling here (line_ocr_engine.py:147-149), an attribute
            # its transformer engine never sets
            raise AttributeError("'TransformerEngineLineOCR' object has no attribute 'net_subsampling'")
        pad = int(self.line_padding_px)
        batches = plan_batches([l.shape[1] for l in lines], self.max_input_horizontal_pixels, self.max_line_width, pad)
        for b in batches:
            if b.max_width + 2 * pad > b.w_batch:
                print_warning(f"WARNING: Line too long for OCR engine. Cropping from {b.max_width + 2 * pad} px "
                              f"down to {b.w_batch}.")
        transcriptions: List[Optional[str]] = [None] * n
        logits_out: List[object] = [None] * n
        coords_out: List[Optional[list]] = [None] * n
        self.recognise_batches(lines, batches, transcriptions, logits_out, coords_out, sparse_logits, no_logits)
        return transcriptions, logits_out, coords_out

    def recognise_batches(self, lines, batches, transcriptions, logits_out, coords_out, sparse_logits=True, no_logits=False):
        """Runs the given reference batches (all of a page, or one rank's share: sharding.ShardedSeq2SeqOCR) and
        fills the three output lists at the positions of their lines."""
        def submit(slot, group):
            images, w_pads, lefts, first = [], [], [], [0]
            for b in group:
                for i, a, e in b.parts:
                    # the reference cuts its batch to w_batch columns BEFORE centring it in 1088 (line_ocr_engine.py:125-127,
                    # transformer_ocr_engine.py:36-40): image columns past w_batch - padding never reach the network
                    images.append(lines[i][:, a:min(e, a + max(b.w_batch - self.line_padding_px, 0))])
                    w_pads.append(b.w_pad)
                    lefts.append(b.pad_left)
                first.append(len(images))
            self._submit(slot, images, w_pads, lefts, first)
            return first

        device_sparse = sparse_logits and not no_logits and self.supports_device_sparsify

        def finish(slot, group, first):
            for b, (labels, logits) in zip(group, self._finish(slot, first, not no_logits, device_sparse)):
                texts = self.decode(labels)
                k = 0
                for i, span in zip(b.line_ids, b.spans):
                    part_logits = logits[k:k + span] if logits is not None else [np.zeros((len(t), 0), np.float32) for t in texts[k:k + span]]
                    # (sparsification is row-wise, so CSC parts built on the GPU merge to the same matrix)
                    text, merged = merge_transcriptions_and_logits(texts[k:k + span], part_logits)
                    k += span
                    transcriptions[i] = text
                    if no_logits:
                        continue
                    coords_out[i] = [0, len(text)]
                    if sparse_logits and not device_sparse:
                        merged = sparse.csc_matrix(np.where(softmax(merged, axis=1) < SPARSE_PROB_THRESHOLD, np.float32(0), merged))
                    logits_out[i] = merged

        # Launches side by side: the encoder of a launch is enqueued by this thread, its decoding loop (blocking, latency-bound:
        # ~31 kernels of 13-40 us per step that leave most of the GPU idle) runs on a worker thread of its slot - the native calls
        # release the GIL - so up to `depth` decoding loops and the next launch's encoder share the GPU (rounds 1-3: one decoding
        # loop at a time next to the next encoder; 2048 lines of 512 px: 7.6 k -> see profiles/r04_s2s_concurrent_decode.txt).
        # Every launch owns its slot's buffers and streams; results land at the positions of their own lines.
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        serial = os.environ.get("POCR_S2S_SERIAL") == "1"        # measurement switch: no overlap between launches
        depth = 1 if serial else max(1, min(int(os.environ.get("POCR_S2S_DEPTH", S2S_DEPTH)), int(getattr(self.net, "num_slots", 2))))
        launches = plan_launches(batches)
        if depth == 1 or len(launches) == 1:
            try:
                for group in launches:
                    finish(0, group, submit(0, group))
            except BaseException:
                self.net.reset()
                raise
            return
        pending = deque()
        pool = ThreadPoolExecutor(max_workers=depth, thread_name_prefix="pocr-s2s")
        try:
            for k, group in enumerate(launches):
                while len(pending) >= depth:                     # slot k % depth is free once launch k - depth has finished
                    pending.popleft().result()
                first = submit(k % depth, group)
                pending.append(pool.submit(finish, k % depth, group, first))
            while pending:
                pending.popleft().result()
        except BaseException:
            for f in pending:                                    # let the running decoding loops end before the engine is reset
                try:
                    f.result()
                except BaseException:
                    pass
            self.net.reset()          # leave the engine usable
            raise
        finally:
            pool.shutdown(wait=True)
