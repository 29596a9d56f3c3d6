"""Multi-GPU sharding of the line-recognition path: one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference is single-device (no torch.distributed / DataParallel anywhere,
SURVEY.md section 2a).  The unit of parallelism is the reference's own chunk
(pero_ocr/ocr_engine/line_ocr_engine.py:79-90): every chunk is an independent forward
pass and a line's logits depend on its chunk's padded width, so whole chunks are dealt
to ranks and never split or re-bucketed.  There is no data-path collective inside the
network; the only exchange is ONE all-gather of the decoded label ids per
`process_lines` call (fixed-stride int32 [lines, T_max] + int32 lengths; <= ~1.2 MB
per 2048 lines, latency-bound on xGMI).  Logits stay on the rank that produced them.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .ocr_engine.line_ocr_engine import Chunk, plan_chunks


def assign_chunks(chunks: Sequence[Chunk], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first on cost = lines * padded width (conv work is linear in
    both).  Returns, per rank, the indices of its chunks (ascending).  Deterministic: every
    rank computes the same assignment from the same widths."""
    return assign_by_cost([len(c.line_ids) * c.w_pad for c in chunks], world_size)


def assign_by_cost(cost: Sequence[int], world_size: int) -> List[List[int]]:
    """LPT: heaviest unit first, each to the currently least loaded rank (ties: lower rank)."""
    order = sorted(range(len(cost)), key=lambda i: (-cost[i], i))
    load = [0] * world_size
    mine: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        mine[r].append(i)
        load[r] += cost[i]
    return [sorted(m) for m in mine]


def _dist():
    import torch.distributed as dist
    return dist


def allgather_labels(labels: np.ndarray, lens: np.ndarray, line_ids: np.ndarray, device=None
                     ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """All-gather ragged label sets.  Each rank passes int32 labels [m_r, T_r], lens [m_r],
    global line ids [m_r] (m_r and T_r may differ per rank, m_r may be 0).  Returns the
    concatenation over ranks (labels padded to the global T_max).  Two collectives:
    a tiny all-gather of (m_r, T_r), then one of the fixed-stride payload."""
    import torch
    dist = _dist()
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    m, t = int(labels.shape[0]), int(labels.shape[1]) if labels.ndim == 2 else 0
    shape = torch.tensor([m, t], dtype=torch.int32, device=dev)
    shapes = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(world)]
    dist.all_gather(shapes, shape)
    shapes = [s.cpu().tolist() for s in shapes]
    m_max = max(s[0] for s in shapes)
    t_max = max(s[1] for s in shapes)
    # payload row = [line_id, len, labels...]; stride t_max + 2
    pay = np.full((m_max, t_max + 2), -1, dtype=np.int32)
    if m:
        pay[:m, 0] = line_ids
        pay[:m, 1] = lens
        pay[:m, 2:2 + t] = labels
    mine = torch.from_numpy(pay).to(dev)
    out = torch.empty((world, m_max, t_max + 2), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out.view(-1), mine.view(-1))      # flat: ncclAllGather layout on RCCL
    out = out.cpu().numpy()
    rows = np.concatenate([out[r, :shapes[r][0]] for r in range(world)], axis=0) if m_max else \
        np.zeros((0, 2), np.int32)
    return rows[:, 2:], rows[:, 1].copy(), rows[:, 0].copy()


class ShardedLineOCR:
    """process_lines over all ranks of the default process group.  Every rank calls it with the
    SAME list of crops (the page stream); each runs the chunks `assign_chunks` gives it on its
    own GPU and all ranks end up with every transcription.

    `recognise(lines, chunk) -> (labels int32 [n, T], lens int32 [n])` is the per-chunk device
    call (PytorchEngineLineOCR on a GPU box; a stand-in in the gloo CPU tests)."""

    def __init__(self, recognise: Callable, characters: Sequence[str], max_input_horizontal_pixels: int,
                 line_padding_px: int = 32, gather_device=None):
        self.recognise = recognise
        self.characters = list(characters)
        self.max_input_horizontal_pixels = max_input_horizontal_pixels
        self.line_padding_px = line_padding_px
        self.gather_device = gather_device

    def process_lines(self, lines) -> List[str]:
        dist = _dist()
        rank, world = dist.get_rank(), dist.get_world_size()
        chunks = plan_chunks([l.shape[1] for l in lines], self.max_input_horizontal_pixels, self.line_padding_px)
        mine = assign_chunks(chunks, world)[rank]
        labs, lens, ids = [], [], []
        t_max = max([chunks[i].frames for i in mine], default=0)
        many = getattr(self.recognise, "many", None)
        results = many(lines, [chunks[ci] for ci in mine]) if many else None     # merged, pipelined launches
        for k, ci in enumerate(mine):
            ch = chunks[ci]
            lab, ln = results[k] if results is not None else self.recognise(lines, ch)
            pad = np.full((lab.shape[0], t_max), -1, dtype=np.int32)
            pad[:, :lab.shape[1]] = lab
            labs.append(pad)
            lens.append(np.asarray(ln, dtype=np.int32))
            ids.append(np.asarray(ch.line_ids, dtype=np.int32))
        if labs:
            L, N, I = np.concatenate(labs), np.concatenate(lens), np.concatenate(ids)
        else:
            L, N, I = np.zeros((0, 0), np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
        gl, gn, gi = allgather_labels(L, N, I, self.gather_device)
        texts: List[Optional[str]] = [None] * len(lines)
        for row, ln, i in zip(gl, gn, gi):
            texts[int(i)] = "".join(self.characters[c] for c in row[:ln])
        return texts


def engine_recogniser(engine) -> Callable:
    """Adapter: PytorchEngineLineOCR -> the `recognise` callable (labels only, no logits)."""
    def recognise(lines, chunk: Chunk):
        pool, offsets, widths = engine._pack_lines(lines, chunk.line_ids)
        engine.model.stage_lines(pool, offsets, widths, chunk.w_pad, engine.line_padding_px)
        _lg, _am, labels, lens = engine.model.run_staged(want_logits=False, want_argmax=False)
        return labels, lens

    def many(lines, chunks):
        """All of this rank's chunks: merged into ragged launches and software-pipelined over the
        engine's two slots (the same path PytorchEngineLineOCR.process_lines uses)."""
        from .ocr_engine.line_ocr_engine import plan_launches
        out = {}

        def finish(launch, handle):
            _kind, slot, _rows, _frames = handle
            _lg, _am, labels, lens = engine.model.slot_collect(slot)
            k = 0
            for ch in launch.chunks:
                m = len(ch.line_ids)
                out[id(ch)] = (labels[k:k + m, :ch.frames], lens[k:k + m])
                k += m

        pending = None
        for j, launch in enumerate(plan_launches(chunks)):
            handle = engine._submit_launch(lines, launch, False, j % 2)
            if pending is not None:
                finish(*pending)
            pending = (launch, handle)
        if pending is not None:
            finish(*pending)
        return [out[id(ch)] for ch in chunks]

    recognise.many = many
    return recognise


class ShardedSeq2SeqOCR:
    """The same scheme for the transformer (sequence-to-sequence) engine: the unit is the reference batch of
    process_lines' "transformer" branch (line_ocr_engine.py:79-119) - a line's result depends on its batch's
    padded width and the decoding loop runs per batch, so whole batches are dealt to ranks (cost = parts x
    padded width).  Transcriptions are exchanged as code points with the all-gather above.

    `recognise(lines, batches) -> {line id: transcription}` runs one rank's batches
    (TransformerEngineLineOCR via `seq2seq_recogniser`; a stand-in in the gloo CPU tests)."""

    def __init__(self, recognise: Callable, max_input_horizontal_pixels: int, max_line_width, line_padding_px: int = 32,
                 gather_device=None):
        self.recognise = recognise
        self.max_input_horizontal_pixels = max_input_horizontal_pixels
        self.max_line_width = max_line_width
        self.line_padding_px = line_padding_px
        self.gather_device = gather_device

    def process_lines(self, lines) -> List[str]:
        from .ocr_engine.transformer_ocr_engine import plan_batches
        dist = _dist()
        rank, world = dist.get_rank(), dist.get_world_size()
        batches = plan_batches([l.shape[1] for l in lines], self.max_input_horizontal_pixels, self.max_line_width,
                               self.line_padding_px)
        mine = assign_by_cost([len(b.parts) * b.w_pad for b in batches], world)[rank]
        texts = self.recognise(lines, [batches[i] for i in mine]) if mine else {}
        ids = sorted(texts)
        t_max = max([len(texts[i]) for i in ids], default=0)
        lab = np.full((len(ids), t_max), -1, dtype=np.int32)
        for k, i in enumerate(ids):
            lab[k, :len(texts[i])] = [ord(ch) for ch in texts[i]]
        lens = np.array([len(texts[i]) for i in ids], dtype=np.int32)
        gl, gn, gi = allgather_labels(lab, lens, np.array(ids, dtype=np.int32), self.gather_device)
        out: List[Optional[str]] = [None] * len(lines)
        for row, ln, i in zip(gl, gn, gi):
            out[int(i)] = "".join(chr(int(c)) for c in row[:ln])
        return out


def seq2seq_recogniser(engine) -> Callable:
    """Adapter: TransformerEngineLineOCR -> the `recognise` callable (transcriptions only)."""
    def recognise(lines, batches):
        n = len(lines)
        texts, lg, co = [None] * n, [None] * n, [None] * n
        engine.recognise_batches(lines, batches, texts, lg, co, sparse_logits=False, no_logits=True)
        return {i: texts[i] for b in batches for i in b.line_ids}
    return recognise
