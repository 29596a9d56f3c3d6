"""Model spec of the layout network ("parsenet_unet64") behind TorchParseNet.get_maps
(pero_ocr/layout_engines/torch_parsenet.py:37-58).

The reference's ParseNet is an opaque TorchScript download that is not in the tree (`torch.jit.load(model_path)`,
torch_parsenet.py:15); what the tree fixes is its contract: uint8 page [h, w, 3] -> zero canvas padded to multiples
of 64 -> float32 * (1/255.) -> `out_map, _ = net(x)` with out_map [1, 5, H, W] at INPUT resolution -> cropped
back to [h, w, 5] (:44-56); channels 0/1 = line heights above / below the baseline, 2 = baseline probability,
3 = line-end probability, 4 = region-border probability (layout_engines/cnn_layout_engine.py:126-196).

This build defines its own network to that contract, like it does for the BiLSTM recogniser: a U-Net whose six
2x2 max-pools are what makes the /64 padding necessary.

  encoder   e0  conv3x3   3 ->  64 + ReLU   @1/1  (fused with the uint8 * (1/255.) staging)      skip x0
            e0p conv3x3  64 ->  64 + ReLU + maxpool 2x2 -> 1/2
            e1  conv3x3  64 -> 128 + ReLU   @1/2                                                   skip x1
            e1p conv3x3 128 -> 128 + ReLU + maxpool -> 1/4
            e2  conv3x3 128 -> 256 + ReLU   @1/4                                                   skip x2
            e2p conv3x3 256 -> 256 + ReLU + maxpool -> 1/8
            e3 / e3p, e4 / e4p, e5 / e5p: 256 -> 256 (skips x3, x4, x5 at 1/8, 1/16, 1/32), e6 @1/64 (bottleneck)
  decoder   y6 = e6;  y_k = ReLU(conv3x3(cat([nearest_up2(y_{k+1}), x_k], channels)))  k = 5 .. 0
            d5, d4, d3: 512 -> 256;  d2: 512 -> 128;  d1: 256 -> 64;  d0: 128 -> 64
  head      conv1x1 64 -> 5;  channels 0, 1: ReLU (heights, >= 0);  channels 2, 3, 4: sigmoid (probabilities)

All convolutions 3x3, stride 1, zero padding 1 (torch.nn.Conv2d semantics), fp32.
Real pero weights cannot be fetched (no network): parity is on seeded synthetic weights (`generate_weights`).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .netspec import uniform01

ARCH = "parsenet_unet64"
PAD_MULTIPLE = 64                      # torch_parsenet.py:44-45
OUT_CHANNELS = 5
# (name, cin, cout, pool)
ENCODER = (
    ("e0", 3, 64, 1), ("e0p", 64, 64, 2),
    ("e1", 64, 128, 1), ("e1p", 128, 128, 2),
    ("e2", 128, 256, 1), ("e2p", 256, 256, 2),
    ("e3", 256, 256, 1), ("e3p", 256, 256, 2),
    ("e4", 256, 256, 1), ("e4p", 256, 256, 2),
    ("e5", 256, 256, 1), ("e5p", 256, 256, 2),
    ("e6", 256, 256, 1),
)
# (name, channels of the up-sampled input, channels of the skip, cout); d_k consumes y_{k+1} and x_k
DECODER = (
    ("d5", 256, 256, 256), ("d4", 256, 256, 256), ("d3", 256, 256, 256),
    ("d2", 256, 256, 128), ("d1", 128, 128, 64), ("d0", 64, 64, 64),
)
HEAD_IN = 64


def tensor_table() -> List[Tuple[str, Tuple[int, ...]]]:
    """Canonical tensor order of the weight blob (the C-ABI contract of pocr_parsenet_create)."""
    t: List[Tuple[str, Tuple[int, ...]]] = []
    for name, cin, cout, _pool in ENCODER:
        t.append((f"{name}.weight", (cout, cin, 3, 3)))
        t.append((f"{name}.bias", (cout,)))
    for name, cup, cskip, cout in DECODER:
        t.append((f"{name}.weight", (cout, cup + cskip, 3, 3)))       # input channels: [up-sampled | skip] (torch.cat order)
        t.append((f"{name}.bias", (cout,)))
    t.append(("head.weight", (OUT_CHANNELS, HEAD_IN, 1, 1)))
    t.append(("head.bias", (OUT_CHANNELS,)))
    return t


def num_weight_floats() -> int:
    return sum(int(np.prod(s)) for _n, s in tensor_table())


def generate_weights(seed: int, head_gain: float = 6.0) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights: He-uniform convs (activations keep their scale through the ReLU stack), small
    biases, a head with `head_gain` / sqrt(fan_in) weights.  The head bias is normally replaced by a data-calibrated
    one (oracle/gen_golden_parsenet.py stores it in the fixture) so that the sigmoid channels are not saturated."""
    out: Dict[str, np.ndarray] = {}
    for ti, (name, shape) in enumerate(tensor_table()):
        n = int(np.prod(shape))
        u = uniform01(seed, 0x9A45 + ti, n)
        if name.endswith(".bias"):
            v = (2.0 * u - 1.0) * 0.05
        elif name.startswith("head."):
            v = (2.0 * u - 1.0) * (head_gain / HEAD_IN ** 0.5)
        else:
            fan_in = shape[1] * 9
            v = (2.0 * u - 1.0) * (6.0 / fan_in) ** 0.5
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def pack_weights(weights: Dict[str, np.ndarray]) -> np.ndarray:
    parts = []
    for name, shape in tensor_table():
        w = np.ascontiguousarray(weights[name], dtype=np.float32)
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {w.shape}")
        parts.append(w.reshape(-1))
    return np.concatenate(parts)


def padded_shape(h: int, w: int) -> Tuple[int, int]:
    """Canvas the network sees (torch_parsenet.py:44-45)."""
    return -(-h // PAD_MULTIPLE) * PAD_MULTIPLE, -(-w // PAD_MULTIPLE) * PAD_MULTIPLE
