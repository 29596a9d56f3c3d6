"""Row softmax used for logit sparsification (counterpart of the reference's
pero_ocr/ocr_engine/softmax.py:4-46: max-subtracted exp, normalised along `axis`,
computed in the input's float32)."""
import numpy as np


def softmax(x, theta=1.0, axis=None):
    y = np.atleast_2d(x)
    if axis is None:
        axis = next(k for k, s in enumerate(y.shape) if s > 1)
    y = y * float(theta)
    y = np.exp(y - np.max(y, axis=axis, keepdims=True))
    p = y / np.sum(y, axis=axis, keepdims=True)
    return p.flatten() if np.ndim(x) == 1 else p
