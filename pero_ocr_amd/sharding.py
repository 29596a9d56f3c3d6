"""Multi-GPU sharding of the line-recognition path: one process per GPU, RCCL over xGMI
through the C ABI (include/pocr.h: pocr_comm_init / pocr_allgather_labels).

The reference is single-device (no torch.distributed / DataParallel anywhere,
SURVEY.md section 2a).  The unit of parallelism is the reference's own chunk
(pero_ocr/ocr_engine/line_ocr_engine.py:79-90): every chunk is an independent forward
pass and a line's logits depend on its chunk's padded width, so whole chunks are dealt
to ranks and never split or re-bucketed.  There is no data-path collective inside the
network; the only exchange is ONE all-gather of the decoded label ids per
`process_lines` call (fixed-stride int32 rows [line id, length, labels...]; <= ~1.2 MB
per 2048 lines, latency-bound on xGMI; the stride and the rows per rank follow from the
chunk plan every rank computes, so no sizes are exchanged).  Logits stay on the rank
that produced them.  The product carries it over RCCL through the C ABI
(pocr_allgather_labels, include/pocr.h); torch.distributed appears only as the carrier
of the CPU tests (TorchDistTransport, "gloo").
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .ocr_engine.line_ocr_engine import Chunk, plan_chunks


def assign_chunks(chunks: Sequence[Chunk], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first on cost = lines * padded width (conv work is linear in
    both).  Returns, per rank, the indices of its chunks (ascending).  Deterministic: every
    rank computes the same assignment from the same widths."""
    sizes = getattr(chunks, "sizes", None)               # a ChunkPlan: costs straight from its arrays
    if sizes is not None:
        return assign_by_cost(sizes * chunks.w_pads, world_size)
    return assign_by_cost([len(c.line_ids) * c.w_pad for c in chunks], world_size)


def assign_by_cost(cost: Sequence[int], world_size: int) -> List[List[int]]:
    """LPT: heaviest unit first, each to the currently least loaded rank (ties: lower rank) - a heap of (load, rank)."""
    import heapq
    c = np.asarray(cost, dtype=np.int64)
    order = np.argsort(-c, kind="stable").tolist()        # heaviest first, ties by index
    cl = c.tolist()
    heap = [(0, r) for r in range(world_size)]
    mine: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        load, r = heap[0]
        mine[r].append(i)
        heapq.heapreplace(heap, (load + cl[i], r))
    for m in mine:
        m.sort()
    return mine


# ---- transports: who carries the one all-gather ------------------------------------------------------------------

class RcclTransport:
    """The product transport: RCCL through the C ABI (pocr_comm_init / pocr_allgather_labels, include/pocr.h) on the
    engine's own GPU - no torch.distributed.  `engine` is a _native.NativeEngine whose comm_init has been called
    (see init_rccl_from_env)."""

    def __init__(self, engine):
        self.engine = engine
        self.rank, self.world = engine.comm_rank, engine.comm_world

    def allgather_i32(self, send: np.ndarray) -> np.ndarray:
        return self.engine.allgather_labels(send)

    def allreduce_max(self, value: float) -> float:
        return self.engine.allreduce_max(value)

    def barrier(self):
        self.engine.allreduce_max(0.0)
        port = getattr(self, "rendezvous_port", None)
        if port is not None:                    # every rank holds its communicator now: the fallback carrier's id file can go
            self.rendezvous_port = None
            comm_rendezvous_cleanup(self.rank, port)


class LocalTransport:
    """A world of one without any communicator (single-GPU runs of the sharded driver)."""
    rank, world = 0, 1

    def allgather_i32(self, send: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(send, dtype=np.int32).reshape(1, -1)

    def allreduce_max(self, value: float) -> float:
        return float(value)

    def barrier(self):
        pass


class TorchDistTransport:
    """torch.distributed's default process group as the carrier: "gloo" in the CPU tests of the N > 1 logic
    (tests/test_sharding.py); not used by the product path."""

    def __init__(self, device=None):
        import torch.distributed as dist
        self.dist, self.device = dist, (device if device is not None else "cpu")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_i32(self, send: np.ndarray) -> np.ndarray:
        import torch
        mine = torch.from_numpy(np.ascontiguousarray(send, dtype=np.int32).reshape(-1)).to(self.device)
        out = torch.empty(self.world * mine.numel(), dtype=torch.int32, device=self.device)
        self.dist.all_gather_into_tensor(out, mine)
        return out.cpu().numpy().reshape(self.world, -1)

    def allreduce_max(self, value: float) -> float:
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self.dist.barrier()


def _id_file(port: int) -> str:
    """Fallback carrier of the unique id on ONE node: a file keyed by the rendezvous port and the launcher's pid (all ranks of a
    torchrun job are children of the same agent process)."""
    import os
    import tempfile
    return os.path.join(tempfile.gettempdir(), f"pocr_rccl_id_{port}_{os.getppid()}")


_ID_MAGIC = b"POCRUID1"


def exchange_unique_id(rank: int, world: int, addr: str, port: int, make_id: Callable[[], bytes], timeout_s: float = 300.0) -> bytes:
    """Out-of-band rendezvous of RCCL's 128-byte unique id: rank 0 creates it and serves it on (addr, port) to the
    other world - 1 ranks, which connect (retrying until rank 0 listens).  If rank 0 cannot bind the port (something else
    owns MASTER_PORT + 1 on this box) it leaves the id in a file instead (`_id_file`: single node, which is all the launchers
    here start); the other ranks look for that file between their connection attempts, so neither side needs to know which
    carrier the other one chose."""
    import os
    import socket
    import time
    if world == 1:
        return make_id()
    path = _id_file(port)
    if rank == 0:
        uid = make_id()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            srv.bind((addr, port))
        except OSError:
            srv.close()
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(_ID_MAGIC + uid)
            os.replace(tmp, path)                 # (atomic: a reader sees nothing or all 136 bytes; removed by comm_rendezvous_cleanup)
            return uid
        srv.listen(world)
        srv.settimeout(timeout_s)
        try:
            for _ in range(world - 1):
                conn, _peer = srv.accept()
                with conn:
                    conn.sendall(uid)
        finally:
            srv.close()
        return uid
    deadline = time.monotonic() + timeout_s
    while True:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == len(_ID_MAGIC) + 128 and blob.startswith(_ID_MAGIC):
                return blob[len(_ID_MAGIC):]
        except OSError:
            pass
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.settimeout(5.0)
                buf = b""
                while len(buf) < 128:
                    part = conn.recv(128 - len(buf))
                    if not part:
                        break
                    buf += part
            if len(buf) == 128:
                return buf
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise RuntimeError(f"rank {rank}: no RCCL unique id from rank 0 at {addr}:{port} (or in {path}) within {timeout_s:.0f} s")
        time.sleep(0.05)


def comm_rendezvous_cleanup(rank: int, port: int) -> None:
    """Rank 0, after the communicator exists on every rank (its first barrier): drop the id file of the fallback carrier."""
    import os
    if rank == 0:
        try:
            os.remove(_id_file(port))
        except OSError:
            pass


def init_rccl_from_env(engine, rank: Optional[int] = None, world: Optional[int] = None) -> RcclTransport:
    """One process per GPU, started by torchrun or bench.py's own launcher: RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT come from the environment.  The id travels over MASTER_PORT + 1 (MASTER_PORT itself belongs to the
    launcher's store; POCR_RDZV_PORT overrides)."""
    import os
    from . import _native
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("POCR_RDZV_PORT", int(os.environ.get("MASTER_PORT", "29533")) + 1))
    uid = exchange_unique_id(rank, world, addr, port, _native.comm_unique_id)
    engine.comm_init(uid, rank, world)
    tr = RcclTransport(engine)
    if world > 1:
        tr.rendezvous_port = port               # (the first barrier's caller may drop the fallback carrier's file: comm_rendezvous_cleanup)
    return tr


ROW_FAILED = -2          # length field of a payload row whose rank could not produce it


def _raise_together(got: np.ndarray, failure, transport, m_of: Sequence[int]) -> None:
    """After the all-gather (padded rows: allgather_rows(compact=False)): if any rank flagged its rows, EVERY rank raises
    (the failing one its own exception)."""
    if failure is not None:
        raise failure
    if got.size and bool(np.any(got[:, 1] == ROW_FAILED)):
        m_max = got.shape[0] // transport.world
        bad = [r for r in range(transport.world) if np.any(got[r * m_max:r * m_max + m_of[r], 1] == ROW_FAILED)]
        raise RuntimeError(f"rank(s) {bad} failed in their share of the page stream; no rank returns a partial result")


def _strings_of_rows(got: np.ndarray, n: int, table) -> List[Optional[str]]:
    """Padded payload rows [line id, length, symbols...] -> the n strings, in one vectorised decode (`table`: the
    characters for label ids, None when the symbols are code points themselves)."""
    from .ocr_engine.pytorch_ocr_engine import labels_to_strings
    if got.shape[0] == 0:
        return [None] * n
    ids = got[:, 0]
    if table is None:
        table = _CodePoints()
    row_of = np.full(n, -1, dtype=np.int64)               # payload row of every line (ids < 0: padding rows)
    valid = np.flatnonzero(ids >= 0)
    row_of[ids[valid]] = valid
    if valid.size == n and bool(np.all(row_of >= 0)):     # the usual case: every line is there - decoded straight into line order
        return labels_to_strings(got, got[:, 1], table, col0=2, rows=row_of)
    texts: List[Optional[str]] = [None] * n
    for i, t in zip(ids[valid].tolist(), labels_to_strings(got, got[:, 1], table, col0=2, rows=valid)):
        texts[i] = t
    return texts


class _CodePoints:
    """`characters` stand-in for rows that already hold code points: labels_to_strings takes its vectorised path with the
    identity table."""
    _cps = None


def allgather_rows(transport, rows: np.ndarray, m_max: int, m_of: Sequence[int], compact: bool = True) -> np.ndarray:
    """ONE fixed-stride all-gather.  rows: int32 [m_r, stride] of this rank; every rank knows m_of (rows per rank) and
    therefore m_max from the shared plan, so no sizes are exchanged.  Returns the ranks' rows back to back; with
    compact=False the padded [world * m_max, stride] array itself (rank r's rows start at r * m_max; padding rows have
    id -1) - the callers below read it in place instead of copying 2 MB per call."""
    stride = rows.shape[1]
    if m_max == 0:
        return np.zeros((0, stride), np.int32)
    if rows.shape[0] == m_max:
        pay = np.ascontiguousarray(rows, dtype=np.int32)
    else:
        pay = np.full((m_max, stride), -1, dtype=np.int32)
        pay[:rows.shape[0]] = rows
    out = transport.allgather_i32(pay.reshape(-1)).reshape(transport.world * m_max, stride)
    if not compact or all(m == m_max for m in m_of):
        return out
    keep = (np.arange(m_max)[None, :] < np.asarray(m_of)[:, None]).reshape(-1)       # one gather instead of a concatenation of slices
    return out[keep]


class ShardedLineOCR:
    """process_lines over all ranks, with the reference's return contract (line_ocr_engine.py:144-177).  Every rank calls
    it with the SAME list of crops (the page stream); each runs the chunks `assign_chunks` gives it on its own GPU.  All
    ranks end up with EVERY transcription (the one all-gather); logits and logit_coords stay on the rank that produced
    them: a rank gets them for the lines of ITS chunks and None elsewhere - which is what the caller that writes them to
    `TextLine` objects needs (page_parser.py:423-430; `PageOCR.process_pages(..., sharded=...)`).

    `recognise(lines, chunk) -> (labels int32 [n, T], lens int32 [n])` is the per-chunk device call of the labels-only
    path (`no_logits=True`); `recognise.full(lines, chunks, sparse_logits, tight_crop_logits) -> (texts, logits, coords)`
    - lists in input order, None outside `chunks` - the one with logits (`engine_recogniser`: PytorchEngineLineOCR on a
    GPU box; stand-ins in the gloo CPU tests).  `transport` carries the all-gather (RcclTransport in the product,
    TorchDistTransport in the CPU tests)."""

    def __init__(self, recognise: Callable, characters: Sequence[str], max_input_horizontal_pixels: int,
                 line_padding_px: int = 32, transport=None):
        self.recognise = recognise
        self.characters = list(characters)
        self.max_input_horizontal_pixels = max_input_horizontal_pixels
        self.line_padding_px = line_padding_px
        self.transport = transport if transport is not None else TorchDistTransport()

    def process_lines(self, lines, sparse_logits=True, tight_crop_logits=False, no_logits=False):
        """-> (transcriptions of ALL lines, logits, logit_coords): the latter two for this rank's lines, None elsewhere
        (all None with no_logits).  ONE collective per call."""
        n = len(lines)
        if no_logits:
            return self._texts_from_labels(lines), [None] * n, [None] * n
        full = getattr(self.recognise, "full", None)
        if full is None:
            raise TypeError("this recogniser returns labels only: call process_lines(..., no_logits=True)")
        tr = self.transport
        chunks = plan_chunks([l.shape[1] for l in lines], self.max_input_horizontal_pixels, self.line_padding_px)
        parts = assign_chunks(chunks, tr.world)
        mine = [chunks[i] for i in parts[tr.rank]]
        m_of = [int(chunks.sizes[p].sum()) for p in parts]
        ids_mine = [i for c in mine for i in c.line_ids]
        # transcriptions travel as code points; a frame emits at most one symbol, so the plan bounds a row's length
        cp_max = max([len(ch) for ch in self.characters], default=1)
        stride = (int((chunks.w_pads.max() // 2) // 2) if len(chunks) else 0) * cp_max + 2
        rows = np.full((m_of[tr.rank], stride), -1, dtype=np.int32)              # [line id, length, code points...]
        logits: List[object] = [None] * n
        coords: List[object] = [None] * n
        failure = None
        try:                                   # (a failing rank still takes part in the collective, with its rows flagged)
            if mine:
                texts_mine, logits, coords = full(lines, mine, sparse_logits, tight_crop_logits)
                for k, i in enumerate(ids_mine):
                    cps = [ord(ch) for ch in texts_mine[i]]
                    if len(cps) > stride - 2:
                        raise RuntimeError(f"line {i}: transcription of {len(cps)} symbols exceeds the plan's bound {stride - 2}")
                    rows[k, 0], rows[k, 1] = i, len(cps)
                    rows[k, 2:2 + len(cps)] = cps
        except Exception as exc:              # noqa: BLE001 - re-raised below, after the collective
            failure = exc
            rows[:, 1] = ROW_FAILED
        got = allgather_rows(tr, rows, max(m_of, default=0), m_of, compact=False)
        _raise_together(got, failure, tr, m_of)
        return _strings_of_rows(got, n, None), logits, coords

    def _texts_from_labels(self, lines) -> List[str]:
        tr = self.transport
        chunks = plan_chunks([l.shape[1] for l in lines], self.max_input_horizontal_pixels, self.line_padding_px)
        parts = assign_chunks(chunks, tr.world)
        mine = parts[tr.rank]
        # payload geometry from the plan alone (identical on every rank): rows per rank, longest label row
        m_of = [int(chunks.sizes[p].sum()) for p in parts]
        t_max = int((chunks.w_pads.max() // 2) // 2) if len(chunks) else 0
        rows = np.full((m_of[tr.rank], t_max + 2), -1, dtype=np.int32)       # [line id, length, labels...]
        # A rank that fails must not leave the others blocked in the collective: it still takes part, with its rows
        # flagged (length = ROW_FAILED), and every rank raises after the exchange.
        failure = None
        try:
            many = getattr(self.recognise, "many", None)
            results = many(lines, [chunks[ci] for ci in mine]) if many else None     # merged, pipelined launches
            k0 = 0
            for k, ci in enumerate(mine):
                ch = chunks[ci]
                lab, ln = results[k] if results is not None else self.recognise(lines, ch)
                m = len(ch.line_ids)
                rows[k0:k0 + m, 0] = ch.line_ids
                rows[k0:k0 + m, 1] = ln
                rows[k0:k0 + m, 2:2 + lab.shape[1]] = lab
                k0 += m
        except Exception as exc:              # noqa: BLE001 - re-raised below, after the collective
            failure = exc
            rows[:, 1] = ROW_FAILED
        got = allgather_rows(tr, rows, max(m_of, default=0), m_of, compact=False)
        _raise_together(got, failure, tr, m_of)
        return _strings_of_rows(got, len(lines), self.characters)


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
        from .ocr_engine.line_ocr_engine import launch_target, plan_launches
        out = {}

        def finish(launch, handle):
            _kind, slot, _rows, _frames = handle
            _lg, _am, labels, lens = engine.model.slot_collect(slot)
            k = 0
            for ch in launch.chunks:
                m = len(ch.line_ids)
                out[id(ch)] = (labels[k:k + m, :ch.frames], lens[k:k + m])
                k += m

        from collections import deque
        pending = deque()
        try:
            # the slots belong to the engine's own queue of launches in flight (process_lines_begin tickets of a page stream may
            # be open): collect those first, so that launch j of this call can take slot j % depth
            while getattr(engine, "_inflight", None):
                engine._collect_oldest()
            from .ocr_engine.line_ocr_engine import pipeline_depth
            depth = pipeline_depth(engine)                # launches in flight, one engine slot each (as process_lines)
            for j, launch in enumerate(plan_launches(chunks, launch_target(engine))):
                while len(pending) >= depth:              # slot j % depth is free once launch j - depth has been collected
                    finish(*pending.popleft())
                pending.append((launch, engine._submit_launch(lines, launch, False, j % depth)))
            while pending:
                finish(*pending.popleft())
        except BaseException:
            engine.model.reset()      # a launch may still be in flight on either slot: leave the engine usable
            raise
        return [out[id(ch)] for ch in chunks]

    def full(lines, chunks, sparse_logits, tight_crop_logits):
        """This rank's chunks with the reference's logits contract: the engine's own process_lines body over them."""
        return engine.process_chunks(lines, chunks, sparse_logits, tight_crop_logits, False)

    recognise.many = many
    recognise.full = full
    return recognise


class ShardedSeq2SeqOCR:
    """The same scheme for the transformer (sequence-to-sequence) engine: the unit is the reference batch of
    process_lines' "transformer" branch (line_ocr_engine.py:79-119) - a line's result depends on its batch's
    padded width and the decoding loop runs per batch, so whole batches are dealt to ranks (cost = parts x
    padded width).  Transcriptions are exchanged as code points with the same single all-gather; the row stride is
    the plan's bound on a transcription's length (every part stops after w_pad / 4 + 1 steps at the latest,
    transformer_ocr_engine.py:74-80).

    `recognise(lines, batches) -> {line id: transcription}` runs one rank's batches
    (TransformerEngineLineOCR via `seq2seq_recogniser`; a stand-in in the gloo CPU tests)."""

    def __init__(self, recognise: Callable, max_input_horizontal_pixels: int, max_line_width, line_padding_px: int = 32,
                 transport=None):
        self.recognise = recognise
        self.max_input_horizontal_pixels = max_input_horizontal_pixels
        self.max_line_width = max_line_width
        self.line_padding_px = line_padding_px
        self.transport = transport if transport is not None else TorchDistTransport()

    def process_lines(self, lines) -> List[str]:
        from .ocr_engine.transformer_ocr_engine import plan_batches
        tr = self.transport
        batches = plan_batches([l.shape[1] for l in lines], self.max_input_horizontal_pixels, self.max_line_width,
                               self.line_padding_px)
        parts = assign_by_cost([len(b.parts) * b.w_pad for b in batches], tr.world)
        mine = parts[tr.rank]
        ids_of = [sorted({i for bi in p for i in batches[bi].line_ids}) for p in parts]
        bound = {}
        for b in batches:                                  # an over-long line is recognised in several parts
            for i, _first, _end in b.parts:
                bound[i] = bound.get(i, 0) + b.w_pad // 4 + 1
        stride = max(bound.values(), default=0) + 2
        m_of = [len(x) for x in ids_of]
        rows = np.full((m_of[tr.rank], stride), -1, dtype=np.int32)
        failure = None
        try:                                   # (a failing rank still takes part in the collective, see ShardedLineOCR)
            texts = self.recognise(lines, [batches[i] for i in mine]) if mine else {}
            ids = sorted(texts)
            if m_of[tr.rank] != len(ids):
                raise RuntimeError("the recogniser did not return every line of this rank's batches")
            for k, i in enumerate(ids):
                cps = [ord(ch) for ch in texts[i]]
                if len(cps) > stride - 2:
                    raise RuntimeError(f"line {i}: transcription of {len(cps)} symbols exceeds the plan's bound {stride - 2}")
                rows[k, 0], rows[k, 1] = i, len(cps)
                rows[k, 2:2 + len(cps)] = cps
        except Exception as exc:              # noqa: BLE001 - re-raised below, after the collective
            failure = exc
            rows[:, 1] = ROW_FAILED
        got = allgather_rows(tr, rows, max(m_of, default=0), m_of, compact=False)
        _raise_together(got, failure, tr, m_of)
        return _strings_of_rows(got, len(lines), None)


def seq2seq_recogniser(engine) -> Callable:
    """Adapter: TransformerEngineLineOCR -> the `recognise` callable (transcriptions only)."""
    def recognise(lines, batches):
        n = len(lines)
        texts, lg, co = [None] * n, [None] * n, [None] * n
        engine.recognise_batches(lines, batches, texts, lg, co, sparse_logits=False, no_logits=True)
        return {i: texts[i] for b in batches for i in b.line_ids}
    return recognise
