"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/model_oracle.py header).

numpy restatement of the reference's host-side algorithm around the network:

  chunk_plan        pero_ocr/ocr_engine/line_ocr_engine.py:79-90   (stable width sort, chunking)
  assemble_batch    line_ocr_engine.py:121-127                      (zero pad, copy at x=32, crop >3840)
  normalise         pero_ocr/ocr_engine/pytorch_ocr_engine.py:61-62 (u8 -> f32 / 255.0, NCHW)
  greedy_ctc        pytorch_ocr_engine.py:13-34 (3-D branch :18-34)
  logit_coords      line_ocr_engine.py:160-163
  softmax           pero_ocr/ocr_engine/softmax.py:4-46
  sparsify          line_ocr_engine.py:168-171
  process_lines     line_ocr_engine.py:57-177 (CTC branch)
  line_confidence   pero_ocr/document_ocr/page_parser.py:485-496 (compute_line_confidence), :437-450 (get_prob),
                    pero_ocr/core/layout.py:65-68 (get_dense_logits); pinned by tests/golden/c1_confidence.json,
                    which oracle/gen_golden_conf.py wrote by executing the reference's own functions

Pinned by: tests/golden/* (outputs of the imported reference, written by
oracle/gen_golden.py) and the CTC known-answer cases of
test/test_decoding/test_decoders.py:24-96 (restated in tests/test_oracle.py).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np
from scipy import sparse

LINE_PADDING_PX = 32       # line_ocr_engine.py:54
NET_SUBSAMPLING = 4        # pytorch_ocr_engine.py:41


def chunk_plan(widths: Sequence[int], max_input_horizontal_pixels: int) -> List[Tuple[List[int], int]]:
    """[(line ids of the chunk, max_width = ceil32(widest))] in processing order."""
    order = [i for i, _w in sorted(enumerate(widths), key=lambda t: -t[1])]   # python sort is stable
    plan = []
    while order:
        max_width = int(np.ceil(widths[order[0]] / 32.0) * 32)
        n = max(1, max_input_horizontal_pixels // max_width)
        plan.append((order[:n], max_width))
        order = order[n:]
    return plan


def assemble_batch(lines: Sequence[np.ndarray], ids: Sequence[int], height: int, max_width: int,
                   max_input_horizontal_pixels: int) -> np.ndarray:
    batch = np.zeros([len(ids), height, max_width + 2 * LINE_PADDING_PX, 3], dtype=np.uint8)
    for row, i in zip(batch, ids):
        row[:, LINE_PADDING_PX:LINE_PADDING_PX + lines[i].shape[1], :] = lines[i]
    if batch.shape[2] > max_input_horizontal_pixels:
        batch = batch[:, :, :max_input_horizontal_pixels]
    return batch


def normalise(batch_u8: np.ndarray) -> np.ndarray:
    return (batch_u8.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2)


def frame_argmax(logits_nct: np.ndarray) -> np.ndarray:
    """First-index argmax over C per frame; NaN counts as maximal (torch.argmax semantics)."""
    x = np.where(np.isnan(logits_nct), np.inf, logits_nct)
    # a NaN must beat +inf that appears earlier: handle by ranking NaN above inf
    has_nan = np.isnan(logits_nct).any(axis=1)
    best = np.argmax(x, axis=1)
    if has_nan.any():
        nan_first = np.argmax(np.isnan(logits_nct), axis=1)
        best = np.where(has_nan, nan_first, best)
    return best.astype(np.int64)


def greedy_ctc(logits_nct: np.ndarray) -> Tuple[np.ndarray, List[np.ndarray]]:
    """-> (per-frame argmax [n,T], list of label-id arrays).  Blank = C-1; a frame is
    dropped if it equals its predecessor (the virtual frame before t=0 is blank) or is blank."""
    n, c, t = logits_nct.shape
    best = frame_argmax(logits_nct)
    prev = np.concatenate([np.full((n, 1), c - 1, dtype=np.int64), best[:, :-1]], axis=1)
    keep = (best != prev) & (best != c - 1)
    return best, [best[i][keep[i]] for i in range(n)]


def labels_to_text(labels: np.ndarray, characters: Sequence[str]) -> str:
    return "".join(characters[int(c)] for c in labels)


def logit_coords(width: int) -> List[int]:
    return [int(LINE_PADDING_PX // NET_SUBSAMPLING), int((LINE_PADDING_PX + width) // NET_SUBSAMPLING)]


def softmax(x: np.ndarray, axis: int) -> np.ndarray:
    y = np.atleast_2d(x) * float(1.0)
    y = y - np.expand_dims(np.max(y, axis=axis), axis)
    y = np.exp(y)
    return y / np.expand_dims(np.sum(y, axis=axis), axis)


def sparsify(line_logits: np.ndarray) -> sparse.csc_matrix:
    """In-place thresholding like the reference (mutates line_logits), then CSC."""
    p = softmax(line_logits, axis=1)
    line_logits[p < 0.0001] = 0
    return sparse.csc_matrix(line_logits)


def process_lines(forward_nct: Callable[[np.ndarray], np.ndarray], lines: Sequence[np.ndarray],
                  characters: Sequence[str], height: int, max_input_horizontal_pixels: int,
                  sparse_logits: bool = True, tight_crop_logits: bool = False, no_logits: bool = False):
    """forward_nct: u8 [n,H,Wpad,3] -> f32 [n,C,T].  Returns (transcriptions, logits,
    logit_coords, extras) with extras = per-line frame argmax + chunk plan (for fixtures)."""
    n = len(lines)
    texts, logits_out, coords = [None] * n, [None] * n, [None] * n
    argmax_out = [None] * n
    plan = chunk_plan([l.shape[1] for l in lines], max_input_horizontal_pixels)
    for ids, max_width in plan:
        batch = assemble_batch(lines, ids, height, max_width, max_input_horizontal_pixels)
        nct = forward_nct(batch)
        best, labels = greedy_ctc(nct)
        ntc = nct.transpose(0, 2, 1)
        for k, i in enumerate(ids):
            texts[i] = labels_to_text(labels[k], characters)
            argmax_out[i] = best[k]
            if no_logits:
                continue
            ll = ntc[k]
            if tight_crop_logits:
                lc = logit_coords(lines[i].shape[1])
                ll = ll[lc[0]:lc[1]]
                coords[i] = [None, None]
            else:
                coords[i] = logit_coords(lines[i].shape[1])
            if sparse_logits:
                ll = sparsify(ll)
            logits_out[i] = ll
    return texts, logits_out, coords, {"frame_argmax": argmax_out, "plan": plan}


def line_confidence(line_logits: sparse.spmatrix, zero_logit_value: float = -80) -> float:
    """Transcription confidence of one line from its SPARSE logits: dropped entries count as -80, the
    per-frame winners are grouped into runs of equal class, a run is worth its highest probability and the
    line is worth its worst run."""
    dense = line_logits.toarray()
    dense[dense == 0] = zero_logit_value
    log_probs = dense - np.logaddexp.reduce(dense, axis=1)[:, None]
    ids = np.argmax(log_probs, axis=1)
    probs = np.exp(np.max(log_probs, axis=1))
    worst, run_best, run_id = 1, 1, -1
    for i, p in zip(ids, probs):
        if i != run_id:
            worst = min(worst, run_best)
            run_id, run_best = i, p
        else:
            run_best = max(run_best, p)
    return min(worst, run_best)
