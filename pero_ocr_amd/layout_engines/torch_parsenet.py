"""Layout network on the MI355X with the reference's TorchParseNet surface.

Drop-in for pero_ocr/layout_engines/torch_parsenet.py:23-103: class name, constructor
`(model_path, device, downsample=4, max_mp=5, detection_threshold=0.2, adaptive_downsample=True)`, the public
attributes the layout engine reads (`last_downsample`, `detection_threshold`, ...), `get_maps(img, downsample)`,
`get_maps_with_optimal_resolution(img)` and `get_med_height(out_map)`.  Where the reference does `torch.jit.load(model_path)`
and `self.net(x)` (:15, :51) this class hands the uint8 page to hand-written HIP kernels through the C ABI
(include/pocr.h, pocr_parsenet_*).  There is no CPU fallback.

`model_path` names a POCRP001 weight blob of this build's "parsenet_unet64" network (pero_ocr_amd/parsenet_spec.py); if the
file does not exist a path of the form "seed:<int>" gives seeded synthetic weights (no real pero checkpoint can be fetched).

cv2.resize(INTER_AREA) of the reference (:42) runs on the device for integer factors (the default downsample 4 and every
integer the adaptive loop lands on); fractional factors - the adaptive second pass - are resampled on the host with the
same area-averaging rule.  Both are restatements of OpenCV's algorithm (cv2 is not installed here): parity unpinned.
"""
from __future__ import annotations

import os
import struct

import numpy as np

from .. import _native, parsenet_spec
from ..ocr_engine.pytorch_ocr_engine import _device_index

MAGIC = b"POCRP001"


def save_blob(path: str, weights) -> None:
    flat = parsenet_spec.pack_weights(weights)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", flat.size))
        f.write(flat.tobytes())


def load_blob(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a POCRP001 layout-network blob")
        (n,) = struct.unpack("<Q", f.read(8))
        flat = np.frombuffer(f.read(), dtype=np.float32)
    if flat.size != n or n != parsenet_spec.num_weight_floats():
        raise ValueError(f"{path}: {flat.size} floats, the network needs {parsenet_spec.num_weight_floats()}")
    return flat


def _area_taps(n_in: int, n_out: int):
    """INTER_AREA tap table of one axis: output o covers the source interval [o * scale, (o + 1) * scale) - at most
    ceil(scale) + 1 taps starting at first[o], each weighted by the covered length, rows normalised (float64).
    -> (weights [n_out, taps], first [n_out]); tap a of output o reads source index min(first[o] + a, n_in - 1)."""
    scale = n_in / n_out
    taps = int(np.ceil(scale)) + 1
    o = np.arange(n_out, dtype=np.float64)
    a, b = o * scale, np.minimum((o + 1.0) * scale, float(n_in))
    first = np.floor(a).astype(np.int64)
    idx = first[:, None] + np.arange(taps, dtype=np.int64)[None, :]
    wgt = np.minimum(b[:, None], idx + 1.0) - np.maximum(a[:, None], idx.astype(np.float64))
    wgt = np.where((idx < n_in) & (wgt > 0), wgt, 0.0)
    wgt /= wgt.sum(axis=1, keepdims=True)
    return wgt, first.astype(np.int32)


def _area_weights(n_in: int, n_out: int):
    """The same table as a sparse [n_out, n_in] matrix (host resample)."""
    from scipy import sparse
    wgt, first = _area_taps(n_in, n_out)
    taps = wgt.shape[1]
    idx = first.astype(np.int64)[:, None] + np.arange(taps, dtype=np.int64)[None, :]
    rows = np.repeat(np.arange(n_out, dtype=np.int64), taps)
    m = sparse.coo_matrix((wgt.reshape(-1), (rows, np.minimum(idx, n_in - 1).reshape(-1))), shape=(n_out, n_in))
    return m.tocsr()


def resize_area(img: np.ndarray, downsample: float) -> np.ndarray:
    """Host INTER_AREA for a fractional factor (the device handles integers): output pixel = area-weighted mean of the
    source rectangle it covers, rounded to nearest; output size cvRound(size / downsample) like cv2.resize(fx=1/ds).
    Separable and sparse (two CSR products, <= ceil(ds) + 1 taps per output pixel): ~0.3 s for a 4k x 3k page - this is
    the path every page after the first takes once get_maps_with_optimal_resolution remembers a fractional factor."""
    h, w = img.shape[:2]
    c = img.shape[2] if img.ndim == 3 else 1
    oh, ow = int(np.rint(h / downsample)), int(np.rint(w / downsample))
    wy, wx = _area_weights(h, oh), _area_weights(w, ow)
    rows = wy @ img.reshape(h, w * c).astype(np.float64)                                   # [oh, w * c]
    cols = wx @ np.ascontiguousarray(rows.reshape(oh, w, c).transpose(1, 0, 2)).reshape(w, oh * c)   # [ow, oh * c]
    out = cols.reshape(ow, oh, c).transpose(1, 0, 2)
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]


class Net(object):
    def __init__(self, model_path, device, max_mp=5):
        self.max_megapixels = max_mp if max_mp is not None else 5
        self.device = device
        if model_path is None:
            self.net = None
            return
        if os.path.exists(model_path):
            flat = load_blob(model_path)
        elif str(model_path).startswith("seed:"):
            flat = parsenet_spec.pack_weights(parsenet_spec.generate_weights(int(str(model_path)[5:])))
        else:
            raise FileNotFoundError(f"layout-network blob {model_path} not found (use 'seed:<int>' for synthetic weights)")
        self.net = _native.NativeParseNet(flat, _device_index(device))


class TorchParseNet(Net):
    def __init__(self, model_path, device, downsample=4, max_mp=5, detection_threshold=0.2, adaptive_downsample=True):
        super().__init__(model_path, device=device, max_mp=max_mp)
        self.detection_threshold = detection_threshold
        self.adaptive_downsample = adaptive_downsample
        self.init_downsample = downsample
        self.last_downsample = downsample
        self.downsample_line_pixel_adapt_threshold = 100
        self.min_line_processing_height = 9
        self.max_line_processing_height = 15
        self.optimal_line_processing_height = 12
        self.min_downsample = 1
        self.max_downsample = 8

    def get_maps(self, img, downsample):
        """ParseNet inference (torch_parsenet.py:37-58): uint8 [H, W, 3] -> float32 [h, w, 5]."""
        ds = float(downsample)
        if ds == int(ds) and ds >= 1:
            return self.net.get_maps(img, int(ds))
        # fractional factor (every page after the first once the adaptive factor is remembered): the area resample runs on
        # the device from the same tap tables the host restatement `resize_area` uses - bit-identical to it (tested)
        im = np.asarray(img)
        oh, ow = int(np.rint(im.shape[0] / ds)), int(np.rint(im.shape[1] / ds))
        (wy, y0), (wx, x0) = _area_taps(im.shape[0], oh), _area_taps(im.shape[1], ow)
        return self.net.get_maps_area(im, wy, y0, wx, x0)

    def get_maps_with_optimal_resolution(self, img):
        """The reference's memory-safe two-pass scheme (:60-93): a first pass at max(last_downsample, megapixel limit);
        if enough line pixels were found and their median height is outside 9..15 px, the factor that would make it
        12 px is adopted (clamped to 1..8 and to the megapixel limit) and, when it differs by more than 20 %, run."""
        limit = np.sqrt((img.shape[0] * img.shape[1]) / (self.max_megapixels * 10e5))
        first = max(self.last_downsample, limit)
        net_downsample = first
        out_map = self.get_maps(img, net_downsample)
        if not self.adaptive_downsample:
            return out_map, net_downsample
        if (out_map[:, :, 2] > self.detection_threshold).sum() > self.downsample_line_pixel_adapt_threshold:
            med_height = self.get_med_height(out_map)
            if med_height > self.max_line_processing_height or med_height < self.min_line_processing_height:
                second = first * (med_height / self.optimal_line_processing_height)
                second = max(min(second, self.max_downsample), self.min_downsample)
                self.last_downsample = second
                second = max(self.last_downsample, limit)
                if second / first < 0.8 or second / first > 1.2:
                    net_downsample = second
                    out_map = self.get_maps(img, net_downsample)
        return out_map, net_downsample

    def get_med_height(self, out_map):
        """Median predicted line height over the detected baseline pixels (:95-103)."""
        heights = (out_map[:, :, 2] > self.detection_threshold).astype(float) * out_map[:, :, 0]
        return np.median(heights[heights > 0])
