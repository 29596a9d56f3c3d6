"""ctypes binding of libpocr_hip.so (C ABI: include/pocr.h).

The HIP library is the product path; there is no CPU fallback.  If the shared
object is missing or no gfx950 device is usable, everything here raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .netspec import NetSpec

LIB_NAME = "libpocr_hip.so"
ABI_VERSION = 13
UNIQUE_ID_BYTES = 128
STAGE_NAMES = ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv8", "conv9",
               "agg", "lstm", "head", "ctc", "total")

_lib: Optional[C.CDLL] = None


class PocrConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("height", C.c_int32), ("num_classes", C.c_int32),
                ("conv_out", C.c_int32), ("lstm_hidden", C.c_int32), ("lstm_layers", C.c_int32),
                ("arch", C.c_int32), ("sa_layers", C.c_int32), ("sa_heads", C.c_int32), ("sa_ff", C.c_int32),
                ("dec_layers", C.c_int32), ("embed_num", C.c_int32)]


class CropSpec(C.Structure):
    """pocr_crop_spec (include/pocr.h): one line of the resident cropper."""
    _fields_ = [("x_min", C.c_double), ("x_max", C.c_double), ("lo", C.c_double), ("hi", C.c_double), ("zoom", C.c_double),
                ("above", C.c_double), ("below", C.c_double), ("rot", C.c_double * 4), ("mode", C.c_int32), ("n_coef", C.c_int32),
                ("coef_off", C.c_int32), ("knot_off", C.c_int32), ("n_x", C.c_int32), ("pad_", C.c_int32)]


CROP_SPEC_DTYPE = np.dtype([("x_min", "<f8"), ("x_max", "<f8"), ("lo", "<f8"), ("hi", "<f8"), ("zoom", "<f8"), ("above", "<f8"),
                            ("below", "<f8"), ("rot", "<f8", (4,)), ("mode", "<i4"), ("n_coef", "<i4"), ("coef_off", "<i4"),
                            ("knot_off", "<i4"), ("n_x", "<i4"), ("pad_", "<i4")])
assert CROP_SPEC_DTYPE.itemsize == C.sizeof(CropSpec)

ARCH_IDS = {"vgg_blstm_ctc": 0, "vgg_sa_ctc": 1, "vgg_sa_s2s": 2}


def make_config(spec: NetSpec) -> "PocrConfig":
    return PocrConfig(ABI_VERSION, spec.height, spec.num_classes, spec.conv_out, spec.lstm_hidden,
                      spec.lstm_layers, ARCH_IDS[spec.arch], spec.sa_layers, spec.sa_heads, spec.sa_ff, spec.dec_layers,
                      spec.embed_num)


# every symbol include/pocr.h declares: name -> (restype, argtypes)
_u8p, _f32p, _i32p, _i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
SYMBOLS = {
    "pocr_num_weight_floats": (C.c_size_t, [C.POINTER(PocrConfig)]),
    "pocr_create": (C.c_int, [C.POINTER(PocrConfig), _f32p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "pocr_destroy": (None, [C.c_void_p]),
    "pocr_last_error": (C.c_char_p, []),
    "pocr_abi_version": (C.c_int, []),
    "pocr_conv_split": (C.c_int, []),
    "pocr_range_fallbacks": (C.c_int64, [C.c_void_p]),
    "pocr_lstm_timeouts": (C.c_int64, [C.c_void_p]),
    "pocr_fallback_ready": (C.c_int, [C.c_void_p, C.c_int32]),
    "pocr_set_embed_id": (C.c_int, [C.c_void_p, C.c_int32]),
    "pocr_device_count": (C.c_int, []),
    "pocr_run_batch": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32, _f32p, _i32p, _i32p, _i32p]),
    "pocr_stage_lines": (C.c_int, [C.c_void_p, _u8p, _i64p, _i32p, C.c_int32, C.c_int32, C.c_int32]),
    "pocr_run_staged": (C.c_int, [C.c_void_p, _f32p, _i32p, _i32p, _i32p]),
    "pocr_ctc_greedy": (C.c_int, [C.c_int, _f32p, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _i32p]),
    "pocr_sparsify": (C.c_int, [C.c_int, _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, _f32p, _i32p, C.c_int64, _i32p, _i64p]),
    "pocr_best_overlap": (C.c_int32, [_i32p, C.c_int32, _i32p, C.c_int32]),
    "pocr_crop_lines": (C.c_int, [C.c_int, _u8p, C.c_int32, C.c_int32, C.c_int32, _f32p, _i64p, _i32p, C.c_int32, C.c_int32, _u8p, _i64p]),
    "pocr_crop_curves": (C.c_int, [C.c_int, _u8p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), _i32p, C.c_int32, C.c_int32, _u8p, _i64p, _f32p]),
    "pocr_cropper_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pocr_cropper_destroy": (None, [C.c_void_p]),
    "pocr_cropper_set_page": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32, C.c_int32]),
    "pocr_cropper_wait_page": (C.c_int, [C.c_void_p]),
    "pocr_cropper_measure": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double),
                                       C.c_int64, _i32p, _i32p]),
    "pocr_cropper_crop": (C.c_int, [C.c_void_p, C.c_int32, _i64p, _u8p, _f32p, _i32p]),
    "pocr_cropper_pinned_crops": (C.c_void_p, [C.c_void_p]),
    "pocr_cropper_crop_resident": (C.c_int, [C.c_void_p, C.c_int32, _i64p, _i32p, C.POINTER(C.c_void_p)]),
    "pocr_crops_release": (None, [C.c_void_p]),
    "pocr_crops_bytes": (C.c_int64, [C.c_void_p]),
    "pocr_crops_read": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _u8p]),
    "pocr_slot_stage_resident": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), _i64p, _i32p, _i32p, C.c_int32, C.c_int32]),
    "pocr_cropper_read_curves": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int64]),
    "pocr_cropper_last_ms": (C.c_float, [C.c_void_p]),
    "pocr_num_slots": (C.c_int, []),
    "pocr_slot_stage_lines": (C.c_int, [C.c_void_p, C.c_int32, _u8p, _i64p, _i32p, C.c_int32, C.c_int32, C.c_int32]),
    "pocr_slot_stage_ragged": (C.c_int, [C.c_void_p, C.c_int32, _u8p, _i64p, _i32p, _i32p, C.c_int32, C.c_int32]),
    "pocr_slot_launch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "pocr_slot_collect": (C.c_int, [C.c_void_p, C.c_int32, _f32p, _i32p, _i32p, _i32p]),
    "pocr_slot_stage_ms": (C.c_int, [C.c_void_p, C.c_int32, _f32p, C.c_int32]),
    "pocr_slot_launch_sparse": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, C.c_float, C.c_int32]),
    "pocr_slot_sparse_nnz": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "pocr_slot_collect_sparse": (C.c_int, [C.c_void_p, C.c_int32, _f32p, _i32p, _i32p, _i64p, _i32p, _i32p, _i32p]),
    "pocr_s2s_stage": (C.c_int, [C.c_void_p, C.c_int32, _u8p, _i64p, _i32p, _i32p, _i32p, C.c_int32]),
    "pocr_s2s_launch": (C.c_int, [C.c_void_p, C.c_int32, _i32p, C.c_int32]),
    "pocr_s2s_decode": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _i32p, _i32p]),
    "pocr_s2s_collect": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _f32p]),
    "pocr_s2s_sparse": (C.c_int, [C.c_void_p, C.c_int32, _i32p, C.c_float, C.POINTER(C.c_int64)]),
    "pocr_s2s_collect_sparse": (C.c_int, [C.c_void_p, C.c_int32, _f32p, _i32p, _i32p, _i64p]),
    "pocr_slot_confidence": (C.c_int, [C.c_void_p, C.c_int32, _f32p]),
    "pocr_slot_reset": (C.c_int, [C.c_void_p, C.c_int32]),
    "pocr_comm_unique_id": (C.c_int, [_u8p]),
    "pocr_comm_init": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32]),
    "pocr_comm_destroy": (C.c_int, [C.c_void_p]),
    "pocr_allgather_labels": (C.c_int, [C.c_void_p, _i32p, C.c_int64, _i32p]),
    "pocr_comm_allreduce_max": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "pocr_comm_info": (C.c_int, [C.c_void_p, _i32p, _i32p]),
    "pocr_device_synchronize": (C.c_int, [C.c_void_p]),
    "pocr_parsenet_num_weight_floats": (C.c_size_t, []),
    "pocr_parsenet_create": (C.c_int, [_f32p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "pocr_parsenet_destroy": (None, [C.c_void_p]),
    "pocr_parsenet_out_shape": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p]),
    "pocr_parsenet_get_maps": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32, C.c_int32, _f32p]),
    "pocr_parsenet_get_maps_area": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), _i32p,
                                              C.c_int32, C.POINTER(C.c_double), _i32p, C.c_int32, _f32p]),
    "pocr_parsenet_last_ms": (C.c_int, [C.c_void_p, _f32p]),
    "pocr_parsenet_range_fallbacks": (C.c_int64, [C.c_void_p]),
    "pocr_last_stage_ms": (C.c_int, [C.c_void_p, _f32p, C.c_int32]),
    "pocr_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "pocr_debug_read": (C.c_int, [C.c_void_p, C.c_int32, _f32p, C.c_size_t, C.POINTER(C.c_size_t)]),
}


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    if lib.pocr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_NAME} ABI {lib.pocr_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def _ptr(arr: Optional[np.ndarray], typ):
    return None if arr is None else arr.ctypes.data_as(typ)


class NativeEngine:
    """Owns one pocr_engine handle (one GPU, one stream)."""

    def __init__(self, spec: NetSpec, flat_weights: np.ndarray, device_id: int = 0):
        self._lib = load()
        self._h = C.c_void_p()
        self.spec = spec
        cfg = make_config(spec)
        w = np.ascontiguousarray(flat_weights, dtype=np.float32)
        rc = self._lib.pocr_create(C.byref(cfg), _ptr(w, _f32p), w.size, int(device_id), C.byref(self._h))
        if rc:
            raise RuntimeError("pocr_create: " + self._err())
        self._n = 0
        self._T = 0
        self.device_id = int(device_id)
        self.num_slots = int(self._lib.pocr_num_slots())
        # per slot: (n, T_max, rows, want_logits, want_argmax, uniform)
        self._slot_shape = [(0, 0, 0, False, False, True)] * self.num_slots

    def _err(self) -> str:
        return (self._lib.pocr_last_error() or b"").decode("utf8", "replace")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pocr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def frames_for(w_pad: int) -> int:
        return (w_pad // 2) // 2

    def _alloc_out(self, n, T, want_logits, want_argmax, rows=None):
        """Uniform chunk (rows None): logits [n,T,C], argmax [n,T].  Ragged: logits [rows,C], argmax [rows]."""
        if rows is None:
            logits = np.empty((n, T, self.spec.num_classes), dtype=np.float32) if want_logits else None
            amax = np.empty((n, T), dtype=np.int32) if want_argmax else None
        else:
            logits = np.empty((rows, self.spec.num_classes), dtype=np.float32) if want_logits else None
            amax = np.empty((rows,), dtype=np.int32) if want_argmax else None
        labels = np.empty((n, T), dtype=np.int32)
        lens = np.empty((n,), dtype=np.int32)
        return logits, amax, labels, lens

    def run_batch(self, batch_u8: np.ndarray, want_logits=True, want_argmax=True):
        """u8 [n,H,w_pad,3] -> (logits [n,T,C] | None, frame_argmax [n,T] | None, labels [n,T], lens [n])"""
        b = np.ascontiguousarray(batch_u8, dtype=np.uint8)
        if b.ndim != 4 or b.shape[1] != self.spec.height or b.shape[3] != 3:
            raise ValueError(f"expected uint8 [n,{self.spec.height},w,3], got {b.shape}")
        n, _, w_pad, _ = b.shape
        T = self.frames_for(w_pad)
        logits, amax, labels, lens = self._alloc_out(n, T, want_logits, want_argmax)
        rc = self._lib.pocr_run_batch(self._h, _ptr(b, _u8p), n, w_pad, _ptr(logits, _f32p), _ptr(amax, _i32p),
                                      _ptr(labels, _i32p), _ptr(lens, _i32p))
        if rc:
            raise RuntimeError("pocr_run_batch: " + self._err())
        self._n, self._T = n, T
        return logits, amax, labels, lens

    def stage_lines(self, pool_u8: np.ndarray, offsets: np.ndarray, widths: np.ndarray, w_pad: int, pad_left: int):
        pool = np.ascontiguousarray(pool_u8, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        wd = np.ascontiguousarray(widths, dtype=np.int32)
        if pool.size == 0:
            pool = np.zeros(1, dtype=np.uint8)
        rc = self._lib.pocr_stage_lines(self._h, _ptr(pool, _u8p), _ptr(off, _i64p), _ptr(wd, _i32p),
                                        int(wd.size), int(w_pad), int(pad_left))
        if rc:
            raise RuntimeError("pocr_stage_lines: " + self._err())
        self._n, self._T = int(wd.size), self.frames_for(w_pad)

    def run_staged(self, want_logits=True, want_argmax=True):
        logits, amax, labels, lens = self._alloc_out(self._n, self._T, want_logits, want_argmax)
        rc = self._lib.pocr_run_staged(self._h, _ptr(logits, _f32p), _ptr(amax, _i32p), _ptr(labels, _i32p),
                                       _ptr(lens, _i32p))
        if rc:
            raise RuntimeError("pocr_run_staged: " + self._err())
        return logits, amax, labels, lens

    # ---- pipelined form: stage -> launch (returns at once) -> collect, per slot -------------
    def slot_stage_lines(self, slot: int, pool_u8, offsets, widths, w_pad: int, pad_left: int):
        pool = np.ascontiguousarray(pool_u8, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        wd = np.ascontiguousarray(widths, dtype=np.int32)
        if pool.size == 0:
            pool = np.zeros(1, dtype=np.uint8)
        rc = self._lib.pocr_slot_stage_lines(self._h, int(slot), _ptr(pool, _u8p), _ptr(off, _i64p), _ptr(wd, _i32p),
                                             int(wd.size), int(w_pad), int(pad_left))
        if rc:
            raise RuntimeError("pocr_slot_stage_lines: " + self._err())
        T = self.frames_for(w_pad)
        self._slot_shape[slot] = (int(wd.size), T, int(wd.size) * T, False, False, True)
        if slot == 0:
            self._n, self._T = int(wd.size), T

    def slot_stage_ragged(self, slot: int, pool_u8, offsets, widths, w_pads, pad_left: int):
        """Lines padded to their own widths w_pads[i] (each line's reference-chunk W_pad)."""
        pool = np.ascontiguousarray(pool_u8, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        wd = np.ascontiguousarray(widths, dtype=np.int32)
        wp = np.ascontiguousarray(w_pads, dtype=np.int32)
        if pool.size == 0:
            pool = np.zeros(1, dtype=np.uint8)
        rc = self._lib.pocr_slot_stage_ragged(self._h, int(slot), _ptr(pool, _u8p), _ptr(off, _i64p), _ptr(wd, _i32p),
                                              _ptr(wp, _i32p), int(wd.size), int(pad_left))
        if rc:
            raise RuntimeError("pocr_slot_stage_ragged: " + self._err())
        frames = (wp // 2) // 2
        self._slot_shape[slot] = (int(wd.size), int(frames.max()), int(frames.sum()), False, False, False)
        return frames

    def slot_stage_resident(self, slot: int, crops, w_pads, pad_left: int):
        """Lines whose crops are already in HBM (LazyCrop objects of ResidentCrops buffers, same device): nothing is copied,
        the first kernel reads them where the cropper wrote them.  The ResidentCrops objects must stay referenced until the
        launch has been collected (the LazyCrop objects do that)."""
        n = len(crops)
        handles = (C.c_void_p * n)(*[c.owner._h for c in crops])
        off = np.array([c.offset for c in crops], dtype=np.int64)
        wd = np.array([c.shape[1] for c in crops], dtype=np.int32)
        wp = np.ascontiguousarray(w_pads, dtype=np.int32)
        if self._lib.pocr_slot_stage_resident(self._h, int(slot), handles, _ptr(off, _i64p), _ptr(wd, _i32p), _ptr(wp, _i32p),
                                              n, int(pad_left)):
            raise RuntimeError("pocr_slot_stage_resident: " + self._err())
        self._slot_resident = getattr(self, "_slot_resident", {})
        self._slot_resident[slot] = list({id(c.owner): c.owner for c in crops}.values())      # keep the buffers alive
        frames = (wp // 2) // 2
        self._slot_shape[slot] = (n, int(frames.max()), int(frames.sum()), False, False, False)
        return frames

    def slot_launch(self, slot: int, want_logits=True, want_argmax=False):
        if self._lib.pocr_slot_launch(self._h, int(slot), 1 if want_logits else 0, 1 if want_argmax else 0):
            raise RuntimeError("pocr_slot_launch: " + self._err())
        n, T, rows, _a, _b, uni = self._slot_shape[slot]
        self._slot_shape[slot] = (n, T, rows, bool(want_logits), bool(want_argmax), uni)

    def slot_collect(self, slot: int):
        n, T, rows, want_logits, want_argmax, uni = self._slot_shape[slot]
        logits, amax, labels, lens = self._alloc_out(n, T, want_logits, want_argmax, None if uni else rows)
        rc = self._lib.pocr_slot_collect(self._h, int(slot), _ptr(logits, _f32p), _ptr(amax, _i32p),
                                         _ptr(labels, _i32p), _ptr(lens, _i32p))
        if rc:
            raise RuntimeError("pocr_slot_collect: " + self._err())
        return logits, amax, labels, lens

    def slot_launch_sparse(self, slot: int, row_begin=None, row_end=None, threshold: float = 1e-4, want_argmax=False):
        """Like slot_launch without dense logits, plus on-device softmax/threshold/CSC compaction."""
        rb = None if row_begin is None else np.ascontiguousarray(row_begin, dtype=np.int32)
        re_ = None if row_end is None else np.ascontiguousarray(row_end, dtype=np.int32)
        if self._lib.pocr_slot_launch_sparse(self._h, int(slot), _ptr(rb, _i32p), _ptr(re_, _i32p), float(threshold),
                                             1 if want_argmax else 0):
            raise RuntimeError("pocr_slot_launch_sparse: " + self._err())
        n, T, rows, _a, _b, uni = self._slot_shape[slot]
        self._slot_shape[slot] = (n, T, rows, False, bool(want_argmax), uni)

    def slot_collect_sparse(self, slot: int):
        """-> (data f32 [nnz], indices i32 [nnz], indptr i32 [n, C+1], line_off i64 [n+1], argmax|None, labels, lens)"""
        n, T, rows, _wl, want_argmax, uni = self._slot_shape[slot]
        total = C.c_int64(0)
        if self._lib.pocr_slot_sparse_nnz(self._h, int(slot), C.byref(total)):
            raise RuntimeError("pocr_slot_sparse_nnz: " + self._err())
        data = np.empty(max(1, total.value), dtype=np.float32)
        indices = np.empty(max(1, total.value), dtype=np.int32)
        indptr = np.empty((n, self.spec.num_classes + 1), dtype=np.int32)
        line_off = np.empty(n + 1, dtype=np.int64)
        _lg, amax, labels, lens = self._alloc_out(n, T, False, want_argmax, None if uni else rows)
        rc = self._lib.pocr_slot_collect_sparse(self._h, int(slot), _ptr(data, _f32p), _ptr(indices, _i32p),
                                                _ptr(indptr, _i32p), _ptr(line_off, _i64p), _ptr(amax, _i32p),
                                                _ptr(labels, _i32p), _ptr(lens, _i32p))
        if rc:
            raise RuntimeError("pocr_slot_collect_sparse: " + self._err())
        return data[:total.value], indices[:total.value], indptr, line_off, amax, labels, lens

    # ---- sequence-to-sequence engine (POCR_ARCH_S2S) -----------------------------------------------
    def s2s_stage(self, slot: int, pool_u8, offsets, widths, w_pads, pad_lefts):
        pool = np.ascontiguousarray(pool_u8, dtype=np.uint8).reshape(-1)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        wd = np.ascontiguousarray(widths, dtype=np.int32)
        wp = np.ascontiguousarray(w_pads, dtype=np.int32)
        pl = np.ascontiguousarray(pad_lefts, dtype=np.int32)
        if pool.size == 0:
            pool = np.zeros(1, dtype=np.uint8)
        if self._lib.pocr_s2s_stage(self._h, int(slot), _ptr(pool, _u8p), _ptr(off, _i64p), _ptr(wd, _i32p),
                                    _ptr(wp, _i32p), _ptr(pl, _i32p), int(wd.size)):
            raise RuntimeError("pocr_s2s_stage: " + self._err())
        self._s2s_n = getattr(self, "_s2s_n", {})
        self._s2s_n[slot] = int(wd.size)

    def s2s_launch(self, slot: int, batch_first):
        """Enqueue the encoder of the staged lines; batch_first [n_batches + 1] = first line of every reference batch."""
        bf = np.ascontiguousarray(batch_first, dtype=np.int32)
        if self._lib.pocr_s2s_launch(self._h, int(slot), _ptr(bf, _i32p), int(bf.size - 1)):
            raise RuntimeError("pocr_s2s_launch: " + self._err())
        self._s2s_nb = getattr(self, "_s2s_nb", {})
        self._s2s_nb[slot] = int(bf.size - 1)

    def s2s_decode(self, slot: int, want_logits: bool = True):
        """Greedy decoding loop (blocking) -> (steps [n_batches], tokens [n, s_max], logits [n, s_max, C] | None)"""
        nb, n = self._s2s_nb[slot], self._s2s_n[slot]
        steps = np.zeros(nb, dtype=np.int32)
        smax = C.c_int32(0)
        if self._lib.pocr_s2s_decode(self._h, int(slot), 1 if want_logits else 0, _ptr(steps, _i32p), C.byref(smax)):
            raise RuntimeError("pocr_s2s_decode: " + self._err())
        tokens = np.empty((n, smax.value), dtype=np.int32)
        logits = np.empty((n, smax.value, self.spec.num_classes), dtype=np.float32) if want_logits else None
        if self._lib.pocr_s2s_collect(self._h, int(slot), _ptr(tokens, _i32p), _ptr(logits, _f32p)):
            raise RuntimeError("pocr_s2s_collect: " + self._err())
        return steps, tokens, logits

    def s2s_sparse(self, slot: int, row_end, threshold: float = 1e-4):
        """CSC triplets of the resident decoder logits, rows [0, row_end[i]) of device line i.
        -> (data [nnz], indices [nnz], indptr [n, C+1], line_off [n+1])"""
        n = self._s2s_n[slot]
        re_ = np.ascontiguousarray(row_end, dtype=np.int32)
        total = C.c_int64(0)
        if self._lib.pocr_s2s_sparse(self._h, int(slot), _ptr(re_, _i32p), float(threshold), C.byref(total)):
            raise RuntimeError("pocr_s2s_sparse: " + self._err())
        data = np.empty(max(1, total.value), dtype=np.float32)
        indices = np.empty(max(1, total.value), dtype=np.int32)
        indptr = np.empty((n, self.spec.num_classes + 1), dtype=np.int32)
        line_off = np.empty(n + 1, dtype=np.int64)
        if self._lib.pocr_s2s_collect_sparse(self._h, int(slot), _ptr(data, _f32p), _ptr(indices, _i32p), _ptr(indptr, _i32p),
                                             _ptr(line_off, _i64p)):
            raise RuntimeError("pocr_s2s_collect_sparse: " + self._err())
        return data[:total.value], indices[:total.value], indptr, line_off

    def slot_confidence(self, slot: int) -> np.ndarray:
        """Per-line transcription confidences of the sparse launch just collected from `slot` (float32 [n])."""
        out = np.empty(self._slot_shape[slot][0], dtype=np.float32)
        if self._lib.pocr_slot_confidence(self._h, int(slot), _ptr(out, _f32p)):
            raise RuntimeError("pocr_slot_confidence: " + self._err())
        return out

    def slot_reset(self, slot: int):
        """Drain and forget whatever `slot` has staged / in flight (error recovery)."""
        if self._lib.pocr_slot_reset(self._h, int(slot)):
            raise RuntimeError("pocr_slot_reset: " + self._err())

    def reset(self):
        """slot_reset on every slot; never raises (used from `finally` blocks)."""
        for sl in range(self.num_slots):
            try:
                self.slot_reset(sl)
            except Exception:
                pass

    # ---- multi-GPU exchange over RCCL (include/pocr.h "multi-GPU exchange") ------------------------
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        if buf.size != UNIQUE_ID_BYTES:
            raise ValueError(f"unique id must be {UNIQUE_ID_BYTES} bytes")
        if self._lib.pocr_comm_init(self._h, _ptr(buf, _u8p), int(rank), int(world)):
            raise RuntimeError("pocr_comm_init: " + self._err())
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_destroy(self):
        if self._lib.pocr_comm_destroy(self._h):
            raise RuntimeError("pocr_comm_destroy: " + self._err())

    def allgather_labels(self, send: np.ndarray) -> np.ndarray:
        """int32 [count] of this rank -> int32 [world, count]: one ncclAllGather (every rank passes the same count)."""
        snd = np.ascontiguousarray(send, dtype=np.int32).reshape(-1)
        out = np.empty((self.comm_world, snd.size), dtype=np.int32)
        if self._lib.pocr_allgather_labels(self._h, _ptr(snd, _i32p), int(snd.size), _ptr(out, _i32p)):
            raise RuntimeError("pocr_allgather_labels: " + self._err())
        return out

    def allreduce_max(self, value: float) -> float:
        v = C.c_double(float(value))
        if self._lib.pocr_comm_allreduce_max(self._h, C.byref(v)):
            raise RuntimeError("pocr_comm_allreduce_max: " + self._err())
        return float(v.value)

    def comm_info(self):
        """(ranks, this rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank); (0, 0) without one."""
        cnt, rk = C.c_int32(0), C.c_int32(0)
        if self._lib.pocr_comm_info(self._h, C.byref(cnt), C.byref(rk)):
            raise RuntimeError("pocr_comm_info: " + self._err())
        return int(cnt.value), int(rk.value)

    def range_fallbacks(self) -> int:
        """Launches re-run on the bf16x3 fall-back engine by the f16x2 range guard (include/pocr.h: pocr_range_fallbacks)."""
        return int(self._lib.pocr_range_fallbacks(self._h))

    def lstm_timeouts(self) -> int:
        """Launches repeated on the step kernels after a hand-off timeout of the resident recurrence (include/pocr.h: pocr_lstm_timeouts)."""
        return int(self._lib.pocr_lstm_timeouts(self._h))

    def fallback_ready(self, wait: bool = False) -> int:
        """1: the bf16x3 fall-back engine of the range guard is there, 0: still being built behind pocr_create (wait=True
        blocks until it is), -1: this engine has none (include/pocr.h: pocr_fallback_ready)."""
        return int(self._lib.pocr_fallback_ready(self._h, 1 if wait else 0))

    def device_synchronize(self):
        if self._lib.pocr_device_synchronize(self._h):
            raise RuntimeError("pocr_device_synchronize: " + self._err())

    def slot_stage_ms(self, slot: int) -> dict:
        buf = np.zeros(len(STAGE_NAMES), dtype=np.float32)
        k = self._lib.pocr_slot_stage_ms(self._h, int(slot), _ptr(buf, _f32p), buf.size)
        return {STAGE_NAMES[i]: float(buf[i]) for i in range(k)}

    def set_embed_id(self, embed_id: int):
        """Row of the style-embeddings table every later launch uses (include/pocr.h pocr_set_embed_id)."""
        if self._lib.pocr_set_embed_id(self._h, int(embed_id)):
            raise RuntimeError("pocr_set_embed_id: " + self._err())

    def set_profiling(self, on: bool):
        self._lib.pocr_set_profiling(self._h, 1 if on else 0)

    def last_stage_ms(self) -> dict:
        buf = np.zeros(len(STAGE_NAMES), dtype=np.float32)
        k = self._lib.pocr_last_stage_ms(self._h, _ptr(buf, _f32p), buf.size)
        return {STAGE_NAMES[i]: float(buf[i]) for i in range(k)}

    def debug_read(self, what: int) -> np.ndarray:
        n = C.c_size_t(0)
        if self._lib.pocr_debug_read(self._h, what, None, 0, C.byref(n)):
            raise RuntimeError("pocr_debug_read: " + self._err())
        out = np.empty(n.value, dtype=np.float32)
        if self._lib.pocr_debug_read(self._h, what, _ptr(out, _f32p), out.size, C.byref(n)):
            raise RuntimeError("pocr_debug_read: " + self._err())
        return out


class NativeParseNet:
    """Owns one pocr_parsenet handle: the layout network on one GPU (include/pocr.h "layout network")."""

    def __init__(self, flat_weights: np.ndarray, device_id: int = 0):
        self._lib = load()
        self._h = C.c_void_p()
        w = np.ascontiguousarray(flat_weights, dtype=np.float32)
        if self._lib.pocr_parsenet_create(_ptr(w, _f32p), w.size, int(device_id), C.byref(self._h)):
            raise RuntimeError("pocr_parsenet_create: " + (self._lib.pocr_last_error() or b"").decode("utf8", "replace"))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pocr_parsenet_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def range_fallbacks(self) -> int:
        """Pages re-run on the bf16x3 fall-back network by the f16x2 range guard (include/pocr.h: pocr_parsenet_range_fallbacks)."""
        return int(self._lib.pocr_parsenet_range_fallbacks(self._h))

    def out_shape(self, h: int, w: int, downsample: int = 1):
        oh, ow = C.c_int32(0), C.c_int32(0)
        if self._lib.pocr_parsenet_out_shape(int(h), int(w), int(downsample), C.byref(oh), C.byref(ow)):
            raise RuntimeError("pocr_parsenet_out_shape: " + (self._lib.pocr_last_error() or b"").decode("utf8", "replace"))
        return oh.value, ow.value

    def get_maps(self, img: np.ndarray, downsample: int = 1) -> np.ndarray:
        """uint8 [H, W, 3] -> float32 [h, w, 5] (area down-sampling by the integer `downsample`, padding, network, crop)."""
        im = np.ascontiguousarray(img, dtype=np.uint8)
        if im.ndim != 3 or im.shape[2] != 3:
            raise ValueError(f"expected a uint8 [H, W, 3] page, got {im.shape}")
        h, w = self.out_shape(im.shape[0], im.shape[1], downsample)
        out = np.empty((h, w, 5), dtype=np.float32)
        if self._lib.pocr_parsenet_get_maps(self._h, _ptr(im, _u8p), im.shape[0], im.shape[1], int(downsample), _ptr(out, _f32p)):
            raise RuntimeError("pocr_parsenet_get_maps: " + (self._lib.pocr_last_error() or b"").decode("utf8", "replace"))
        return out

    def get_maps_area(self, img: np.ndarray, wy, y0, wx, x0) -> np.ndarray:
        """Fractional INTER_AREA on the device from separable tap tables (wy [oh, ty], y0 [oh], wx [ow, tx], x0 [ow]),
        then the network: uint8 [H, W, 3] -> float32 [oh, ow, 5]."""
        im = np.ascontiguousarray(img, dtype=np.uint8)
        wy_, wx_ = np.ascontiguousarray(wy, dtype=np.float64), np.ascontiguousarray(wx, dtype=np.float64)
        y0_, x0_ = np.ascontiguousarray(y0, dtype=np.int32), np.ascontiguousarray(x0, dtype=np.int32)
        oh, ow = wy_.shape[0], wx_.shape[0]
        out = np.empty((oh, ow, 5), dtype=np.float32)
        dp = C.POINTER(C.c_double)
        if self._lib.pocr_parsenet_get_maps_area(self._h, _ptr(im, _u8p), im.shape[0], im.shape[1], oh, ow, wy_.ctypes.data_as(dp),
                                                 _ptr(y0_, _i32p), wy_.shape[1], wx_.ctypes.data_as(dp), _ptr(x0_, _i32p), wx_.shape[1],
                                                 _ptr(out, _f32p)):
            raise RuntimeError("pocr_parsenet_get_maps_area: " + (self._lib.pocr_last_error() or b"").decode("utf8", "replace"))
        return out

    def last_ms(self) -> float:
        ms = C.c_float(0)
        self._lib.pocr_parsenet_last_ms(self._h, C.byref(ms))
        return float(ms.value)


class ResidentCrops:
    """One device buffer of crops that stayed in HBM (pocr_cropper_crop_resident): owns the pocr_crops handle."""

    def __init__(self, lib, handle, device_id: int):
        self._lib, self._h, self.device_id = lib, handle, int(device_id)
        self.nbytes = int(lib.pocr_crops_bytes(handle))

    def read(self, offset: int, nbytes: int) -> np.ndarray:
        out = np.empty(int(nbytes), dtype=np.uint8)
        if nbytes and self._lib.pocr_crops_read(self._h, int(offset), int(nbytes), _ptr(out, _u8p)):
            raise RuntimeError("pocr_crops_read: " + (self._lib.pocr_last_error() or b"").decode("utf8", "replace"))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pocr_crops_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LazyCrop:
    """A text-line crop `uint8 [H, w, C]` that lives in HBM (one line of a ResidentCrops buffer).  It quacks like the numpy
    array the reference puts into `line.crop` (shape / ndim / dtype / size, np.asarray, indexing, copy) and is copied to the
    host only when somebody actually looks at the pixels; PytorchEngineLineOCR.process_lines stages such crops on the GPU
    directly (pocr_slot_stage_resident), so between the cropper and the recogniser nothing crosses PCIe."""
    dtype = np.dtype(np.uint8)
    ndim = 3

    def __init__(self, owner: ResidentCrops, offset: int, shape):
        self.owner, self.offset, self.shape = owner, int(offset), tuple(int(v) for v in shape)
        self._host = None

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def nbytes(self):
        return self.size

    def materialise(self) -> np.ndarray:
        if self._host is None:
            self._host = self.owner.read(self.offset, self.size).reshape(self.shape)
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.materialise()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        return self.materialise()[key]

    def __len__(self):
        return self.shape[0]

    def copy(self):
        return self.materialise().copy()

    def astype(self, dtype, **kw):
        return self.materialise().astype(dtype, **kw)


class NativeCropper:
    """Owns one pocr_cropper handle: the resident line cropper of one GPU (include/pocr.h "resident line cropper")."""

    def __init__(self, device_id: int = 0):
        self._lib = load()
        self._h = C.c_void_p()
        self._page = None
        self._n = 0
        self._widths = None
        if self._lib.pocr_cropper_create(int(device_id), C.byref(self._h)):
            raise RuntimeError("pocr_cropper_create: " + self._err())

    def _err(self) -> str:
        return (self._lib.pocr_last_error() or b"").decode("utf8", "replace")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pocr_cropper_destroy(self._h)
            self._h = C.c_void_p()
            self._page = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_page(self, page: np.ndarray):
        """Starts the upload of a uint8 [H, W(, C)] page and returns; the page stays resident for every later crop."""
        img = np.ascontiguousarray(page, dtype=np.uint8)
        if img.ndim == 2:
            img = img[:, :, None]
        if img.ndim != 3:
            raise ValueError(f"expected a [H, W, C] page, got {img.shape}")
        if self._lib.pocr_cropper_set_page(self._h, _ptr(img, _u8p), img.shape[0], img.shape[1], img.shape[2]):
            raise RuntimeError("pocr_cropper_set_page: " + self._err())
        self._page = img                   # keeps the buffer alive while the helper thread reads it
        self.channels = int(img.shape[2])

    def wait_page(self):
        if self._lib.pocr_cropper_wait_page(self._h):
            raise RuntimeError("pocr_cropper_wait_page: " + self._err())

    def measure(self, specs: np.ndarray, knots: np.ndarray, coefs: np.ndarray):
        """specs: CROP_SPEC_DTYPE [n]; -> (widths int32 [n], status int32 [n])."""
        sp = np.ascontiguousarray(specs, dtype=CROP_SPEC_DTYPE)
        kn = np.ascontiguousarray(knots, dtype=np.float64)
        cf = np.ascontiguousarray(coefs, dtype=np.float64)
        n = int(sp.shape[0])
        widths, status = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        dp = C.POINTER(C.c_double)
        if self._lib.pocr_cropper_measure(self._h, sp.ctypes.data_as(C.c_void_p), n, kn.ctypes.data_as(dp), kn.size, cf.ctypes.data_as(dp),
                                          cf.size, _ptr(widths, _i32p), _ptr(status, _i32p)):
            raise RuntimeError("pocr_cropper_measure: " + self._err())
        self._n, self._widths, self._status0 = n, widths, status.copy()
        return widths, status

    def crop_resident(self, line_height: int, device_id: int):
        """Crops of the lines measured last, left in HBM -> (list of LazyCrop (None where the line failed), status)."""
        n, widths = self._n, self._widths
        ch = self.channels
        sizes = widths.astype(np.int64) * int(line_height) * ch
        crop_off = np.zeros(n, dtype=np.int64)
        np.cumsum(sizes[:-1], out=crop_off[1:])
        status = np.zeros(n, dtype=np.int32)
        h = C.c_void_p()
        if self._lib.pocr_cropper_crop_resident(self._h, int(line_height), _ptr(crop_off, _i64p), _ptr(status, _i32p), C.byref(h)):
            raise RuntimeError("pocr_cropper_crop_resident: " + self._err())
        owner = ResidentCrops(self._lib, h, device_id)
        out = [LazyCrop(owner, crop_off[i], (line_height, int(widths[i]), ch)) if status[i] == 0 and widths[i] > 0 else None
               for i in range(n)]
        return out, status

    def crop(self, line_height: int, copy: bool = True, want_grids: bool = False):
        """Crops of the lines measured last -> (list of uint8 [line_height, w_i, C] (None where the line failed), status[, grids]).
        copy=False: the arrays are views of the cropper's pinned buffer, valid until the next crop call."""
        n, widths = self._n, self._widths
        ch = self.channels
        sizes = widths.astype(np.int64) * int(line_height) * ch
        crop_off = np.zeros(n, dtype=np.int64)
        np.cumsum(sizes[:-1], out=crop_off[1:])
        total = int(sizes.sum())
        status = np.zeros(n, dtype=np.int32)
        grid = np.zeros(max(1, 2 * total // ch), dtype=np.float32) if want_grids else None
        if self._lib.pocr_cropper_crop(self._h, int(line_height), _ptr(crop_off, _i64p), None, _ptr(grid, _f32p), _ptr(status, _i32p)):
            raise RuntimeError("pocr_cropper_crop: " + self._err())
        flat = None
        if total:
            addr = self._lib.pocr_cropper_pinned_crops(self._h)
            flat = np.ctypeslib.as_array(C.cast(addr, _u8p), shape=(total,))
            if copy:
                flat = flat.copy()
        crops = []
        for i in range(n):
            if status[i] or widths[i] == 0:
                crops.append(None)
            else:
                o = int(crop_off[i])
                crops.append(flat[o:o + int(sizes[i])].reshape(int(line_height), int(widths[i]), ch))
        if not want_grids:
            return crops, status
        grids, g = [], 0
        for i in range(n):
            if self._status0[i] or widths[i] == 0:
                grids.append(None)
                continue
            m = 2 * int(line_height) * int(widths[i])          # (a line that failed in the column kernel keeps its slot)
            grids.append(None if status[i] else grid[g:g + m].reshape(int(line_height), int(widths[i]), 2).copy())
            g += m
        return crops, status, grids

    def read_curves(self):
        """Test hook: float64 [4, w_i] per line measured with status 0 (None otherwise), after crop()."""
        total = int(4 * self._widths[self._status0 == 0].astype(np.int64).sum())
        buf = np.zeros(max(1, total), dtype=np.float64)
        if self._lib.pocr_cropper_read_curves(self._h, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size):
            raise RuntimeError("pocr_cropper_read_curves: " + self._err())
        out, o = [], 0
        for i in range(self._n):
            if self._status0[i]:
                out.append(None)
                continue
            w = int(self._widths[i])
            out.append(buf[o:o + 4 * w].reshape(4, w).copy())
            o += 4 * w
        return out

    def last_ms(self) -> float:
        return float(self._lib.pocr_cropper_last_ms(self._h))


def ctc_greedy(logits_ntc: np.ndarray, device_id: int = 0):
    """float32 [n, T, C] -> (frame_argmax [n, T], labels [n, T] (-1 padded), lens [n]) on the GPU."""
    lib = load()
    x = np.ascontiguousarray(logits_ntc, dtype=np.float32)
    n, T, Cc = x.shape
    amax = np.empty((n, T), np.int32)
    labels = np.empty((n, T), np.int32)
    lens = np.empty(n, np.int32)
    if lib.pocr_ctc_greedy(int(device_id), _ptr(x, _f32p), n, T, Cc, _ptr(amax, _i32p), _ptr(labels, _i32p), _ptr(lens, _i32p)):
        raise RuntimeError("pocr_ctc_greedy: " + (lib.pocr_last_error() or b"").decode("utf8", "replace"))
    return amax, labels, lens


def sparsify(logits_ntc: np.ndarray, threshold: float = 1e-4, device_id: int = 0):
    """float32 [n, T, C] -> list of n scipy csc_matrix [T, C]: softmax, p < threshold -> 0, CSC - on the GPU."""
    from scipy import sparse
    lib = load()
    x = np.ascontiguousarray(logits_ntc, dtype=np.float32)
    n, T, Cc = x.shape
    cap = max(1, x.size)
    data, indices = np.empty(cap, np.float32), np.empty(cap, np.int32)
    indptr, line_off = np.empty((n, Cc + 1), np.int32), np.empty(n + 1, np.int64)
    if lib.pocr_sparsify(int(device_id), _ptr(x, _f32p), n, T, Cc, float(threshold), _ptr(data, _f32p), _ptr(indices, _i32p), cap,
                         _ptr(indptr, _i32p), _ptr(line_off, _i64p)):
        raise RuntimeError("pocr_sparsify: " + (lib.pocr_last_error() or b"").decode("utf8", "replace"))
    return [sparse.csc_matrix((data[line_off[i]:line_off[i + 1]].copy(), indices[line_off[i]:line_off[i + 1]].copy(), indptr[i]),
                              shape=(T, Cc)) for i in range(n)]


def best_overlap(text1, text2) -> int:
    """find_best_overlap (line_ocr_engine.py:196-211) in native code; works on str or sequences of hashable symbols."""
    lib = load()
    codes = {}
    a = np.array([codes.setdefault(ch, len(codes)) for ch in text1], dtype=np.int32)
    b = np.array([codes.setdefault(ch, len(codes)) for ch in text2], dtype=np.int32)
    if a.size == 0 or b.size == 0:
        return 0
    return int(lib.pocr_best_overlap(_ptr(a, _i32p), int(a.size), _ptr(b, _i32p), int(b.size)))


def crop_lines(page: np.ndarray, grids, device_id: int = 0):
    """page uint8 [H, W, C]; grids: list of float32 [line_h, w_i, 2] (x, y) -> list of uint8 [line_h, w_i, C] crops
    (cv2.remap INTER_LINEAR / BORDER_CONSTANT arithmetic) in one GPU call."""
    lib = load()
    img = np.ascontiguousarray(page, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    H, W, Cc = img.shape
    if not grids:
        return []
    line_h = int(grids[0].shape[0])
    flat = [np.ascontiguousarray(g, dtype=np.float32).reshape(-1) for g in grids]
    widths = np.array([g.shape[1] for g in grids], dtype=np.int32)
    coord_off = np.concatenate([[0], np.cumsum([f.size for f in flat])[:-1]]).astype(np.int64)
    sizes = widths.astype(np.int64) * line_h * Cc
    crop_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    coords = np.concatenate(flat) if flat else np.zeros(0, np.float32)
    if coords.size == 0:
        coords = np.zeros(2, np.float32)
    out = np.zeros(max(1, int(sizes.sum())), dtype=np.uint8)
    if lib.pocr_crop_lines(int(device_id), _ptr(img, _u8p), H, W, Cc, _ptr(coords, _f32p), _ptr(coord_off, _i64p), _ptr(widths, _i32p),
                           int(widths.size), line_h, _ptr(out, _u8p), _ptr(crop_off, _i64p)):
        raise RuntimeError("pocr_crop_lines: " + (lib.pocr_last_error() or b"").decode("utf8", "replace"))
    return [out[int(o):int(o + s)].reshape(line_h, int(w), Cc).copy() for o, s, w in zip(crop_off, sizes, widths)]


def crop_curves(page: np.ndarray, curves, rows, rots, device_id: int = 0, want_grids: bool = False):
    """page uint8 [H, W, C]; per line: curves float64 [4, w_i] (base_x, base_y, normal_x, normal_y), rows float64 [line_h],
    rots float64 [2, 2] -> crops uint8 [line_h, w_i, C] (+ the float32 grids if want_grids): grid generation + remap on the GPU."""
    lib = load()
    img = np.ascontiguousarray(page, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    H, W, Cc = img.shape
    if not curves:
        return ([], []) if want_grids else []
    line_h = int(len(rows[0]))
    widths = np.array([c.shape[1] for c in curves], dtype=np.int32)
    cv = np.concatenate([np.ascontiguousarray(c, dtype=np.float64).reshape(-1) for c in curves] + [np.zeros(1)])
    rw = np.ascontiguousarray(np.stack(rows), dtype=np.float64)
    rt = np.ascontiguousarray(np.stack(rots), dtype=np.float64).reshape(-1)
    sizes = widths.astype(np.int64) * line_h * Cc
    crop_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    out = np.zeros(max(1, int(sizes.sum())), dtype=np.uint8)
    gsz = widths.astype(np.int64) * line_h * 2
    grid = np.zeros(max(1, int(gsz.sum())), dtype=np.float32) if want_grids else None
    dp = C.POINTER(C.c_double)
    if lib.pocr_crop_curves(int(device_id), _ptr(img, _u8p), H, W, Cc, cv.ctypes.data_as(dp), rw.ctypes.data_as(dp), rt.ctypes.data_as(dp),
                            _ptr(widths, _i32p), int(widths.size), line_h, _ptr(out, _u8p), _ptr(crop_off, _i64p), _ptr(grid, _f32p)):
        raise RuntimeError("pocr_crop_curves: " + (lib.pocr_last_error() or b"").decode("utf8", "replace"))
    crops = [out[int(o):int(o + s)].reshape(line_h, int(w), Cc).copy() for o, s, w in zip(crop_off, sizes, widths)]
    if not want_grids:
        return crops
    goff = np.concatenate([[0], np.cumsum(gsz)[:-1]])
    return crops, [grid[int(o):int(o + s)].reshape(line_h, int(w), 2).copy() for o, s, w in zip(goff, gsz, widths)]


def comm_unique_id() -> bytes:
    """Rank 0: the 128-byte RCCL rendezvous id (ncclGetUniqueId) to hand to every rank's comm_init."""
    buf = np.zeros(UNIQUE_ID_BYTES, dtype=np.uint8)
    lib = load()
    if lib.pocr_comm_unique_id(_ptr(buf, _u8p)):
        raise RuntimeError("pocr_comm_unique_id: " + (lib.pocr_last_error() or b"").decode("utf8", "replace"))
    return buf.tobytes()


def device_count() -> int:
    return int(load().pocr_device_count())


def conv_split() -> int:
    """Arithmetic of the conv / GEMM kernels of this process: 2 = f16x2 (default), 3 = bf16x3, 0 = fp32 MFMA."""
    return int(load().pocr_conv_split())
