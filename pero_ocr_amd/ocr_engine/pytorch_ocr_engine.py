"""MI355X-native CTC line-recognition engine with the reference's engine surface.

Drop-in for pero_ocr/ocr_engine/pytorch_ocr_engine.py: class name, constructor
`(json_def, device, batch_size=8)`, attributes and `run_ocr(batch_data) ->
(decoded strings, float32 [n, T, C] logits)` follow PytorchEngineLineOCR (:37-74);
`process_lines` comes from BaseEngineLineOCR.  Where the reference does
`torch.jit.load(checkpoint)` + `model(x)` + `greedy_decode_ctc` (:52-57, :59-74,
:13-34) this engine hands the uint8 crops to hand-written HIP kernels through the
C ABI in include/pocr.h (ctypes).  There is no CPU fallback: constructing the
engine without a usable gfx950 device raises.

Engine JSON: the reference's keys (line_px_height, line_vertical_scale, checkpoint,
characters, net_name, optional embed_*/max_line_width) plus the build-specific key
  "net": {"arch": "vgg_blstm_ctc", "conv_out": 512, "lstm_hidden": 256, "lstm_layers": 2,
          "weight_seed": <int, optional>}
  or, for the self-attention variant (LineSelfAttentionEncoder, transformer.py:366-385; the JSON names of
  transformer.build_net :13-20 are accepted as aliases):
  "net": {"arch": "vgg_sa_ctc", "conv_out": 512, "sa_layers"|"encoder_layers": 2, "sa_heads"|"heads": 8,
          "sa_ff"|"dim_ff": 2048}
`checkpoint` names a POCRW001 weight blob (pero_ocr_amd/netspec.py); if the file does
not exist and "weight_seed" is given, seeded synthetic weights are generated instead
(there is no network access to fetch real pero checkpoints).
"""
from __future__ import annotations

import os
import threading
from typing import List, Tuple

import numpy as np

from .. import _native, netspec
from scipy import sparse

from .line_ocr_engine import BaseEngineLineOCR, Chunk, SPARSE_PROB_THRESHOLD

BLANK_PLACEHOLDER = "\u200B"      # pytorch_ocr_engine.py:42


def _device_index(device) -> int:
    if isinstance(device, int):
        return device
    dtype = getattr(device, "type", None)
    if dtype is None:
        s = str(device)
        dtype, _, idx = s.partition(":")
        index = int(idx) if idx else 0
    else:
        index = getattr(device, "index", None)
        index = 0 if index is None else int(index)
    if dtype == "cpu":
        raise RuntimeError("pero_ocr_amd has no CPU path: the engine runs on an MI355X (device 'cuda:<i>'). "
                           "Use the reference engine for CPU inference.")
    return index


_CP_CACHE = {}


def labels_to_strings(labels: np.ndarray, lens: np.ndarray, characters, col0: int = 0, rows=None) -> List[str]:
    """Label ids -> strings (pytorch_ocr_engine.py:29-32: ''.join(chars[c] for c in line)).  When every entry of `characters`
    is ONE code point (the reference's character sets) all lines are decoded in one pass - gather the code points of every
    kept label, one UTF-32 decode, then slice per line - 20x less host time than a Python loop per symbol (a 2048-line page
    stream: 10 ms -> 0.5 ms per rank, which matters once 8 ranks share the GPU work).  `col0`: the symbols of a row start at
    that column, and `rows` (int array) picks and orders the rows to decode (the gathered payload rows of sharding.py are read in
    place, in line order)."""
    n = labels.shape[0] if rows is None else len(rows)
    if n == 0:
        return []
    if getattr(characters, "_cps", 0) is None:              # sharding._CodePoints: the rows hold code points already
        cps = None
        identity = True
    else:
        identity = False
        key = id(characters)
        ent = _CP_CACHE.get(key)
        if ent is None or ent[0] is not characters:
            cps = np.array([ord(c) for c in characters], dtype=np.uint32) if all(isinstance(c, str) and len(c) == 1 for c in characters) else None
            if len(_CP_CACHE) > 16:
                _CP_CACHE.clear()
            ent = _CP_CACHE[key] = (characters, cps)
        cps = ent[1]
    width = max(int(labels.shape[1]) - int(col0), 0)        # symbols a row can hold: a length beyond it is clipped, never read from the next row
    if cps is None and not identity:
        pick = range(n) if rows is None else [int(r) for r in rows]
        return ["".join(characters[c] for c in labels[i, col0:col0 + min(max(int(lens[i]), 0), width)]) for i in pick]
    ln = np.minimum(np.maximum(np.asarray(lens, dtype=np.int64), 0), width)
    row_id = np.arange(n, dtype=np.int64) if rows is None else np.asarray(rows, dtype=np.int64)
    if rows is not None:
        ln = ln[row_id]
    ends = np.cumsum(ln)
    total = int(ends[-1])
    if total == 0:                                          # nothing but empty lines (or rows of zero columns)
        return [""] * n
    flat_labels = (labels if labels.flags.c_contiguous else np.ascontiguousarray(labels)).reshape(-1)
    sym_of = (lambda c: chr(int(c))) if identity else (lambda c: characters[c])
    if flat_labels.size >= 2 ** 31 or total + n >= 2 ** 31:
        return ["".join(sym_of(c) for c in labels[int(i), col0:col0 + int(k)]) for i, k in zip(row_id, ln)]
    # Work proportional to the SYMBOLS, not to lines x the widest row (a gathered page stream is 2048 rows of 272 columns holding
    # ~30 symbols each), 32-bit indices, np.take: the output is the page's symbols with one NUL behind every line - position q
    # belongs to line line_of[q] and is read at that row's start + its offset in the line -, then one decode and one split
    if cps is not None and cps.size and int(cps.min()) == 0:
        text = cps[flat_labels[(np.repeat(row_id * labels.shape[1] + col0 - (ends - ln), ln) + np.arange(total)).astype(np.int64)]].astype("<u4").tobytes().decode("utf-32-le")
        return [text[e - k:e] for e, k in zip(ends.tolist(), ln.tolist())]         # (a character set with NUL in it: slice per line)
    ln1 = ln + 1
    ends1 = ends + np.arange(1, n + 1)
    line_of = np.repeat(np.arange(n, dtype=np.int32), ln1)
    src = np.take((row_id * labels.shape[1] + col0 - (ends1 - ln1)).astype(np.int32), line_of)
    src += np.arange(total + n, dtype=np.int32)
    stops = ends1 - 1                                       # where the NULs go
    src[stops] = 0
    sym = np.take(flat_labels, src)
    sym = sym.astype("<u4") if identity else np.take(cps, sym, mode="clip")
    if identity:                                            # rows of code points may hold a NUL themselves: then slice per line, as above
        sym[stops] = 1
        if not bool(np.all(sym)):
            sym[stops] = 0
            text = sym.astype("<u4", copy=False).tobytes().decode("utf-32-le")
            return [text[e - k - 1:e - 1] for e, k in zip(ends1.tolist(), ln.tolist())]
    sym[stops] = 0
    return sym.astype("<u4", copy=False).tobytes().decode("utf-32-le").split("\x00")[:n]


def greedy_decode_ctc(scores_probs, chars, device_id: int = 0) -> List[str]:
    """GPU counterpart of the reference's module-level greedy_decode_ctc (pytorch_ocr_engine.py:13-34,
    3-D branch): scores_probs [N, C, T], blank is the last class; returns the decoded strings."""
    x = np.asarray(scores_probs, dtype=np.float32)
    if x.ndim != 3:
        raise ValueError("scores_probs must be [N, C, T]")
    _amax, labels, lens = _native.ctc_greedy(np.ascontiguousarray(x.transpose(0, 2, 1)), device_id)
    return labels_to_strings(labels, lens, chars)


class _HostTensor:
    """numpy array behind the three calls callers chain on a parameter: `.cpu().detach().numpy()`."""

    def __init__(self, array):
        self._a = np.array(array, dtype=np.float32)
        self.shape = self._a.shape

    def cpu(self):
        return self

    def detach(self):
        return self

    def numpy(self):
        return self._a


class _EmbeddingView:
    """Stand-in for the TorchScript `Embedding` sub-module of an embed_id model: `original_name`, `weight`, `parameters()`."""
    original_name = "Embedding"

    def __init__(self, weight):
        self.weight = _HostTensor(weight)

    def parameters(self):
        return iter([self.weight])


_FAST_CSC = None          # None: not probed yet; dict / False: the attribute-level construction below is / is not in use
_FAST_CSC_SCIPY = ((1, 8), (1, 17))      # scipy versions [lo, hi) on which the attribute-level construction has been run against the constructor (1.15.3 here)
_FAST_CSC_CHECKED = 0     # matrices of this process so far: the first 64 and then every 256th have their GPU triplets verified to be canonical
_FAST_CSC_SHAPES = set()  # ... and the first matrix of every distinct number of rows (a new kernel path - long lines, the fall-back engine - shows up as one)
_FAST_CSC_LOCK = threading.Lock()


def _canonical_triplets(indices, indptr, shape) -> bool:
    """What csc_matrix calls canonical format: int32 index arrays, indptr non-decreasing from 0 to nnz, row indices strictly
    increasing inside every column (sorted, no duplicates) and inside [0, rows)."""
    if indices.dtype != np.int32 or indptr.dtype != np.int32 or len(indptr) != shape[1] + 1:
        return False
    if indptr[0] != 0 or indptr[-1] != len(indices) or np.any(np.diff(indptr) < 0):
        return False
    if len(indices) and (indices.min() < 0 or indices.max() >= shape[0]):
        return False
    inner = np.ones(len(indices), bool)
    inner[indptr[:-1][indptr[:-1] < len(indices)]] = False          # first entry of every non-empty column
    return bool(np.all(np.diff(indices, prepend=-1)[inner] > 0)) if len(indices) else True


def _probe_fast_csc(data, indices, indptr, shape):
    """-> the per-instance state of a constructor-built csc_matrix if the attribute-level construction of _csc_from_device gives
    an object indistinguishable from the constructor's on this scipy, else False."""
    env = os.environ.get("POCR_FAST_CSC", "")
    import scipy
    ver = tuple(int(x) for x in scipy.__version__.split(".")[:2] if x.isdigit())
    if not (env == "1" or (env != "0" and _FAST_CSC_SCIPY[0] <= ver < _FAST_CSC_SCIPY[1])):
        return False
    try:
        ref = sparse.csc_matrix((data, indices, indptr), shape=shape)
        ref.sort_indices(); ref.sum_duplicates()                   # what the constructor leaves to be found out lazily
        state = {k: v for k, v in ref.__dict__.items() if k not in ("data", "indices", "indptr", "_shape")}
        probe = sparse.csc_matrix.__new__(sparse.csc_matrix)
        probe.__dict__.update(state)
        probe.data, probe.indices, probe.indptr, probe._shape = data, indices, indptr, tuple(int(v) for v in shape)
        fresh = sparse.csc_matrix((data, indices, indptr), shape=shape)      # an independent constructor-built object to compare with
        extra = set(vars(probe)) - set(vars(fresh))             # only the two lazily cached flags may be new
        if (set(vars(fresh)) <= set(vars(probe)) and extra <= {"_has_sorted_indices", "_has_canonical_format"} and
                probe.shape == fresh.shape and probe.nnz == fresh.nnz and
                (probe != fresh).nnz == 0 and np.array_equal(probe.toarray(), fresh.toarray()) and
                np.array_equal((probe.T @ probe).toarray(), (fresh.T @ fresh).toarray())):
            return state
    except Exception:
        pass
    return False


def _csc_from_device(data, indices, indptr, shape):
    """scipy.sparse.csc_matrix((data, indices, indptr), shape) for the triplets the GPU's compaction builds (sorted row
    indices, no duplicates, int32 index arrays).  The public constructor spends ~16 us per matrix on validation - 4 ms per
    256-line launch, a third of the launch's GPU time - so the object is assembled attribute by attribute instead: on the
    scipy versions this was run against (_FAST_CSC_SCIPY; POCR_FAST_CSC=0 turns it off, =1 forces it), with the per-instance
    state taken from a matrix the CONSTRUCTOR built, and only the flags the contract of the GPU compaction implies -
    sorted indices, canonical format - are asserted, after that contract has been verified on the first matrices of the
    process (_canonical_triplets).  Everything else goes through the constructor."""
    global _FAST_CSC, _FAST_CSC_CHECKED
    if _FAST_CSC is None:
        with _FAST_CSC_LOCK:         # (decoding loops of the sequence-to-sequence engine build matrices on worker threads: one probe)
            if _FAST_CSC is None:
                _FAST_CSC = _probe_fast_csc(data, indices, indptr, shape)
    state = _FAST_CSC               # one snapshot: another thread may switch the fast path off between the check and the use
    if not state:
        return sparse.csc_matrix((data, indices, indptr), shape=shape)
    # the GPU compaction's contract, verified where it is cheap and where a new path would first show: the first matrices of the
    # process, the first of every row count, and a sample of the rest
    _FAST_CSC_CHECKED += 1
    rows_key = int(shape[0])
    if _FAST_CSC_CHECKED <= 64 or _FAST_CSC_CHECKED % 256 == 0 or rows_key not in _FAST_CSC_SHAPES:
        if len(_FAST_CSC_SHAPES) < 4096:
            _FAST_CSC_SHAPES.add(rows_key)
        if not _canonical_triplets(indices, indptr, shape):
            _FAST_CSC = False
            return sparse.csc_matrix((data, indices, indptr), shape=shape)
    m = sparse.csc_matrix.__new__(sparse.csc_matrix)
    m.__dict__.update(state)
    m.data, m.indices, m.indptr, m._shape = data, indices, indptr, (int(shape[0]), int(shape[1]))
    return m


class PytorchEngineLineOCR(BaseEngineLineOCR):
    def __init__(self, json_def, device, batch_size=8):
        super().__init__(json_def, device, batch_size=batch_size)
        self.net_subsampling = 4
        self.characters = list(self.characters) + [BLANK_PLACEHOLDER]
        self._load_exported_model()
        # pytorch_ocr_engine.py:46-50: "mean" = the last row of the model's embeddings table
        if self.embed_id == "mean":
            self.embed_id = self.get_mean_embed_id()
        if self.embed_id is not None:
            self.embed_id = self.embed_id                   # (the setter hands the row to the device now that the model exists)
        elif self.net_spec.embed_num:
            raise ValueError("the model has an embeddings layer but the engine JSON carries no embed_id "
                             "(the reference's model call fails without the ids argument)")

    # `embed_id` stays live-writable like in the reference, where run_ocr reads it on every call (:64-66) and
    # user_scripts/select_embed_id.py:80 assigns it between process_lines calls
    @property
    def embed_id(self):
        return self._embed_id

    @embed_id.setter
    def embed_id(self, value):
        self._embed_id = value
        model = getattr(self, "model", None)
        if model is not None and value is not None and value != "mean":
            model.set_embed_id(int(value))

    def get_mean_embed_id(self):
        if not self.net_spec.embed_num:
            raise ValueError('embed_id "mean": the model has no embeddings layer')
        return self.net_spec.embed_num                      # embeddings_layer.weight.shape[0] - 1

    # reference name kept (pytorch_ocr_engine.py:52)
    def _load_exported_model(self):
        net_cfg = dict(self.config.get("net", {}))
        if os.path.exists(self.checkpoint):
            spec, weights = netspec.load_blob(self.checkpoint)
        elif "weight_seed" in net_cfg:
            spec = netspec.NetSpec(num_classes=len(self.characters), height=int(self.line_px_height),
                                   conv_out=int(net_cfg.get("conv_out", 512)),
                                   lstm_hidden=int(net_cfg.get("lstm_hidden", 256)),
                                   lstm_layers=int(net_cfg.get("lstm_layers", 2)),
                                   arch=net_cfg.get("arch", netspec.ARCH),
                                   sa_layers=int(net_cfg.get("sa_layers", net_cfg.get("encoder_layers", 2))),
                                   sa_heads=int(net_cfg.get("sa_heads", net_cfg.get("heads", 8))),
                                   sa_ff=int(net_cfg.get("sa_ff", net_cfg.get("dim_ff", 2048))),
                                   embed_num=int(self.embed_num or 0))
            weights = netspec.generate_weights(spec, int(net_cfg["weight_seed"]))
        else:
            raise FileNotFoundError(f"weight blob {self.checkpoint} not found and no net.weight_seed in the engine JSON")
        if spec.num_classes != len(self.characters):
            raise ValueError(f"model has {spec.num_classes} classes, engine JSON implies {len(self.characters)} "
                             "(characters + blank)")
        if spec.height != int(self.line_px_height):
            raise ValueError(f"model height {spec.height} != line_px_height {self.line_px_height}")
        self.net_spec = spec
        self.model = _native.NativeEngine(spec, netspec.pack_weights(spec, weights), _device_index(self.device))
        # the one thing callers ask the model object itself: the embeddings table (user_scripts/select_embed_id.py:114-120
        # walks `model.named_modules()` for the module called "embeddings_layer" and clusters its parameters)
        modules = [("", self.model)]
        if spec.embed_num:
            modules.append(("embeddings_layer", _EmbeddingView(weights["embeddings_layer.weight"])))
        self.model.named_modules = lambda: iter(modules)

    def run_ocr(self, batch_data) -> Tuple[List[str], np.ndarray]:
        """uint8 [n, H, W_pad, 3] -> (decoded strings, float32 logits [n, T, C])."""
        logits, _amax, labels, lens = self.model.run_batch(batch_data, want_logits=True, want_argmax=False)
        return labels_to_strings(labels, lens, self.characters), logits

    def _pack_lines(self, lines, line_ids):
        flat = [np.ascontiguousarray(lines[i], dtype=np.uint8).reshape(-1) for i in line_ids]
        widths = np.array([lines[i].shape[1] for i in line_ids], dtype=np.int32)
        sizes = np.array([f.size for f in flat], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        pool = np.concatenate(flat) if flat else np.zeros(0, np.uint8)
        return pool, offsets, widths

    def _submit_launch(self, lines, launch, want_logits: bool, slot: int, sparse_rows=None):
        """Ragged, asynchronous: the crops of one or more reference chunks go to the GPU un-padded, each
        line with the padded width of its own chunk (the zero padding of line_ocr_engine.py:121-123
        happens inside the first kernel's staging); the call returns as soon as the work is enqueued on
        the slot's streams.  sparse_rows = None: dense logits (if wanted); (row_begin, row_end) or
        (None, None): the softmax / p < 1e-4 / CSC step of line_ocr_engine.py:168-171 runs on the GPU."""
        ids = launch.line_ids
        mine = [lines[i] for i in ids]
        dev = getattr(self.model, "device_id", None)
        if all(isinstance(c, _native.LazyCrop) and c._host is None and c.owner.device_id == dev and c.shape[2] == 3 for c in mine):
            # crops the resident cropper left in HBM on this GPU: staged in place (reference: the same numpy arrays go from
            # LineCropper to PageOCR, page_parser.py:384-393 -> 418-430)
            frames = self.model.slot_stage_resident(slot, mine, launch.w_pads, self.line_padding_px)
        else:
            pool, offsets, widths = self._pack_lines(lines, ids)
            frames = self.model.slot_stage_ragged(slot, pool, offsets, widths, launch.w_pads, self.line_padding_px)
        if sparse_rows is not None and want_logits:
            self.model.slot_launch_sparse(slot, sparse_rows[0], sparse_rows[1], SPARSE_PROB_THRESHOLD)
            return ("sparse", slot, sparse_rows, frames)
        self.model.slot_launch(slot, want_logits=want_logits, want_argmax=False)
        return ("dense", slot, None, frames)

    def _collect_launch(self, handle):
        """-> (strings, per-line logits: list of [T_i, C] views / csc matrices, or None)"""
        kind, slot, rows, frames = handle
        if kind == "dense":
            logits, _amax, labels, lens = self.model.slot_collect(slot)
            per_line = None
            if logits is not None:
                ends = np.cumsum(frames)
                per_line = [logits[e - f:e] for e, f in zip(ends, frames)]
            return labels_to_strings(labels, lens, self.characters), per_line
        data, indices, indptr, line_off, _amax, labels, lens = self.model.slot_collect_sparse(slot)
        conf = self.model.slot_confidence(slot)          # page_parser.py:485-496 on the same kept set, computed on the GPU
        n, C = indptr.shape[0], indptr.shape[1] - 1
        mats = []
        for i in range(n):
            a, b = int(line_off[i]), int(line_off[i + 1])
            nrows = (int(rows[1][i]) - int(rows[0][i])) if rows[0] is not None else int(frames[i])
            mats.append(_csc_from_device(data[a:b], indices[a:b], indptr[i], (nrows, C)))
        return labels_to_strings(labels, lens, self.characters), mats, conf

    def _recognise_chunk(self, lines, chunk: Chunk, want_logits: bool):
        """One reference chunk, blocking (the seam BaseEngineLineOCR falls back to)."""
        from .line_ocr_engine import Launch
        texts, per_line = self._collect_launch(self._submit_launch(lines, Launch([chunk]), want_logits, 0))[:2]
        return texts, (np.stack(per_line) if per_line is not None else None)

    supports_device_sparsify = True
    device_sparsify_max_frames = 1 << 30   # csrc/sparsify.hpp works on blocks of 64 frames: no limit on a line's length (rounds 1-3: 1024)

    def frame_argmax(self, batch_data) -> np.ndarray:
        """Per-frame class ids [n, T] (what greedy_decode_ctc's torch.argmax sees)."""
        _l, amax, _lab, _len = self.model.run_batch(batch_data, want_logits=False, want_argmax=True)
        return amax
