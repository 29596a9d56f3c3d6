"""Numerically stable soft-max over one axis (host side).

Counterpart of the helper the reference uses before thresholding logits
(pero_ocr/ocr_engine/softmax.py:4-46, called at line_ocr_engine.py:169): same arguments
(`theta` multiplier, `axis`, default = first axis longer than one), same float32-in/float32-out
behaviour.  On the GPU path this work is done by csrc/sparsify.hpp; this function serves the
host fallback of `process_lines` (engines without device sparsification) and the tests.
"""
import numpy as np


def _default_axis(shape):
    for k, extent in enumerate(shape):
        if extent > 1:
            return k
    raise StopIteration("softmax: no axis longer than one")


def softmax(x, theta=1.0, axis=None):
    arr = np.asarray(x)
    work = arr.reshape(1, -1) if arr.ndim == 1 else arr
    if axis is None:
        axis = _default_axis(work.shape)
    scaled = work * float(theta)
    shifted = scaled - scaled.max(axis=axis, keepdims=True)
    e = np.exp(shifted)
    out = e / e.sum(axis=axis, keepdims=True)
    return out.reshape(arr.shape) if arr.ndim == 1 else out
