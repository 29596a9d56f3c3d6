"""Model spec, weight-blob format and deterministic weight generator for the
line recogniser this package runs.

Why this file exists: the reference's CTC recogniser is an opaque TorchScript
blob loaded at pero_ocr/ocr_engine/pytorch_ocr_engine.py:52-57; its architecture is
not in the reference tree.  The only network source in the tree is
pero_ocr/ocr_engine/transformer.py, whose VGG backbone
(create_vgg_block_2d :51-72, VGG_conv_module :75-148, ConvolutionalEncoder :335-363,
instantiated with subsampling=(8,4), layers_2d=17, base_channels=64, conv_blocks=4)
is the topology used here.  The BiLSTM and the CTC head follow torch.nn.LSTM /
torch.nn.Linear semantics.  I/O contract = pytorch_ocr_engine.py:12,59-74:
float32 [N,3,H,W] in (uint8/255, BGR), float32 [N,C,T] out, T = W/4, blank = last class.

Topology ("vgg_blstm_ctc", H = line height, W = padded width):

  conv1   3->64   3x3 p1  ReLU
  conv2  64->64   3x3 p1  ReLU   maxpool (2,2)   -> H/2 x W/2
  conv3  64->128  3x3 p1  ReLU
  conv4 128->128  3x3 p1  ReLU   maxpool (2,2)   -> H/4 x W/4
  conv5 128->256  3x3 p1  ReLU
  conv6 256->256  3x3 p1  ReLU
  conv7 256->256  3x3 p1  ReLU   maxpool (2,1)   -> H/8 x W/4
  conv8 256->512  3x3 p1  LeakyReLU(0.01)
  conv9 512->512  3x3 p1  LeakyReLU(0.01)  (identity pool)  BatchNorm2d(512, eval, eps 1e-5)
  agg   512->E    (H/8)x1 valid  LeakyReLU(0.01)            -> [N,E,T]
  L x BiLSTM(hidden Hh)  (gates i,f,g,o; b_ih + b_hh)       -> [N,T,2Hh]
  head  Linear(2Hh -> C)                                     -> [N,C,T]

Weights cannot be committed (21 M parameters), so they come from a counter-based
generator (splitmix64) that is pure integer arithmetic + one float scale, hence
identical on every machine.
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

import numpy as np

ARCH = "vgg_blstm_ctc"
ARCH_SA = "vgg_sa_ctc"          # same backbone, self-attention encoder instead of the BiLSTM (BASELINE config 4)
ARCH_S2S = "vgg_sa_s2s"         # backbone + self-attention encoder + autoregressive transformer decoder
                                # (TransformerOCR, pero_ocr/ocr_engine/transformer.py:388-508)
LN_EPS = 1e-5
MAGIC = b"POCRW001"
LEAKY_SLOPE = 0.01
BN_EPS = 1e-5

# (cin, cout, activation, pool(h,w) applied after the activation)
CONV_PLAN: Tuple[Tuple[int, int, str, Tuple[int, int]], ...] = (
    (3, 64, "relu", (1, 1)),
    (64, 64, "relu", (2, 2)),
    (64, 128, "relu", (1, 1)),
    (128, 128, "relu", (2, 2)),
    (128, 256, "relu", (1, 1)),
    (256, 256, "relu", (1, 1)),
    (256, 256, "relu", (2, 1)),
    (256, 512, "leaky", (1, 1)),
    (512, 512, "leaky", (1, 1)),
)
NET_SUBSAMPLING_W = 4
NET_SUBSAMPLING_H = 8


@dataclass(frozen=True)
class NetSpec:
    num_classes: int            # C, blank included (blank = C-1)
    height: int = 40            # line_px_height; must be a multiple of 8
    in_channels: int = 3
    conv_out: int = 512         # E, aggregation conv output channels
    lstm_hidden: int = 256
    lstm_layers: int = 2
    arch: str = ARCH
    # "vgg_sa_ctc" only: self-attention encoder (LineSelfAttentionEncoder, transformer.py:366-385)
    sa_layers: int = 2
    sa_heads: int = 8
    sa_ff: int = 2048
    # "vgg_sa_s2s" only: decoder layers (same width / heads / feed-forward size as the encoder,
    # transformer.build_net :13-47); num_classes then counts the symbols + sentence boundary + ignore
    dec_layers: int = 2
    # style embeddings (pytorch_ocr_engine.py:46-50, 64-66: `model(batch, ids)` with `model.embeddings_layer`): the
    # reference's models with an embeddings layer are opaque TorchScript, so - like the BiLSTM and the head - this build
    # defines its own: Embedding(embed_num + 1, 2E) -> per-channel (scale, shift) of the aggregated features [N, E, T]
    # before the sequence layers, f * (1 + s) + b.  The last row is the "mean" embedding (get_mean_embed_id, :49-50).
    # 0 = the model has no embeddings layer.
    embed_num: int = 0

    def __post_init__(self):
        if self.arch not in (ARCH, ARCH_SA, ARCH_S2S):
            raise ValueError(f"unknown arch {self.arch!r}")
        if self.arch == ARCH_S2S and self.dec_layers < 1:
            raise ValueError("dec_layers must be >= 1")
        if self.arch in (ARCH_SA, ARCH_S2S):
            if self.conv_out % self.sa_heads or (self.conv_out // self.sa_heads) % 16:
                raise ValueError("conv_out / sa_heads must be a multiple of 16")
            if self.sa_ff % 16 or self.sa_layers < 1:
                raise ValueError("sa_ff must be a multiple of 16 and sa_layers >= 1")
        if self.height % 8 or self.height <= 0:
            raise ValueError("height must be a positive multiple of 8")
        if self.conv_out % 16 or self.lstm_hidden % 16:
            raise ValueError("conv_out and lstm_hidden must be multiples of 16")
        if self.in_channels != 3:
            raise ValueError("in_channels must be 3 (BGR crops)")
        if self.embed_num < 0 or (self.embed_num and self.arch == ARCH_S2S):
            raise ValueError("embed_num must be >= 0 (and 0 for the seq2seq engine)")

    @property
    def agg_height(self) -> int:
        return self.height // NET_SUBSAMPLING_H

    def to_json(self) -> dict:
        return asdict(self)

    @staticmethod
    def from_json(d: dict) -> "NetSpec":
        return NetSpec(**{k: d[k] for k in
                          ("num_classes", "height", "in_channels", "conv_out",
                           "lstm_hidden", "lstm_layers", "arch", "sa_layers", "sa_heads", "sa_ff", "dec_layers", "embed_num")
                          if k in d})


def tensor_table(spec: NetSpec) -> List[Tuple[str, Tuple[int, ...], str, int]]:
    """Canonical tensor order of the weight blob: (name, shape, kind, fan_in).
    Shapes follow PyTorch conventions (Conv2d [Cout,Cin,kh,kw], LSTM [4H,In],
    Linear [out,in]).  This order is the C-ABI contract of pocr_create()."""
    t: List[Tuple[str, Tuple[int, ...], str, int]] = []
    for i, (cin, cout, _act, _pool) in enumerate(CONV_PLAN, start=1):
        t.append((f"conv{i}.weight", (cout, cin, 3, 3), "conv_w", cin * 9))
        t.append((f"conv{i}.bias", (cout,), "bias", cin * 9))
    c_last = CONV_PLAN[-1][1]
    t.append(("bn.gamma", (c_last,), "bn_gamma", 0))
    t.append(("bn.beta", (c_last,), "bn_beta", 0))
    t.append(("bn.mean", (c_last,), "bn_mean", 0))
    t.append(("bn.var", (c_last,), "bn_var", 0))
    ah = spec.agg_height
    t.append(("agg.weight", (spec.conv_out, c_last, ah, 1), "conv_w", c_last * ah))
    t.append(("agg.bias", (spec.conv_out,), "bias", c_last * ah))
    if spec.arch in (ARCH_SA, ARCH_S2S):
        e, ff = spec.conv_out, spec.sa_ff
        t.append(("sa.norm.weight", (e,), "ln_w", 0))
        t.append(("sa.norm.bias", (e,), "ln_b", 0))
        for l in range(spec.sa_layers):
            t.append((f"sa{l}.in_proj.weight", (3 * e, e), "sa_w", e))
            t.append((f"sa{l}.in_proj.bias", (3 * e,), "sa_b", e))
            t.append((f"sa{l}.out_proj.weight", (e, e), "sa_w", e))
            t.append((f"sa{l}.out_proj.bias", (e,), "sa_b", e))
            t.append((f"sa{l}.lin1.weight", (ff, e), "sa_w", e))
            t.append((f"sa{l}.lin1.bias", (ff,), "sa_b", e))
            t.append((f"sa{l}.lin2.weight", (e, ff), "sa_w", ff))
            t.append((f"sa{l}.lin2.bias", (e,), "sa_b", ff))
            t.append((f"sa{l}.norm1.weight", (e,), "ln_w", 0))
            t.append((f"sa{l}.norm1.bias", (e,), "ln_b", 0))
            t.append((f"sa{l}.norm2.weight", (e,), "ln_w", 0))
            t.append((f"sa{l}.norm2.bias", (e,), "ln_b", 0))
        if spec.arch == ARCH_S2S:
            # decoder layer = DecoderLayer (transformer.py:388-463): cached self-attention, attention over the
            # encoder output, ReLU feed-forward; post-norm.  Then Embedding(C, E) and Linear(E, C) (:497-498).
            for l in range(spec.dec_layers):
                for att in ("self", "cross"):
                    t.append((f"dec{l}.{att}.in_proj.weight", (3 * e, e), "sa_w", e))
                    t.append((f"dec{l}.{att}.in_proj.bias", (3 * e,), "sa_b", e))
                    t.append((f"dec{l}.{att}.out_proj.weight", (e, e), "sa_w", e))
                    t.append((f"dec{l}.{att}.out_proj.bias", (e,), "sa_b", e))
                t.append((f"dec{l}.lin1.weight", (ff, e), "sa_w", e))
                t.append((f"dec{l}.lin1.bias", (ff,), "sa_b", e))
                t.append((f"dec{l}.lin2.weight", (e, ff), "sa_w", ff))
                t.append((f"dec{l}.lin2.bias", (e,), "sa_b", ff))
                for k in (1, 2, 3):
                    t.append((f"dec{l}.norm{k}.weight", (e,), "ln_w", 0))
                    t.append((f"dec{l}.norm{k}.bias", (e,), "ln_b", 0))
            t.append(("dec.embed.weight", (spec.num_classes, e), "embed", e))
            t.append(("dec.out.weight", (spec.num_classes, e), "head_w", e))
            t.append(("dec.out.bias", (spec.num_classes,), "s2s_b", e))
            return t
        t.append(("head.weight", (spec.num_classes, e), "head_w", e))
        t.append(("head.bias", (spec.num_classes,), "head_b", e))
        if spec.embed_num:
            t.append(("embeddings_layer.weight", (spec.embed_num + 1, 2 * e), "style_embed", e))
        return t
    hh = spec.lstm_hidden
    for l in range(spec.lstm_layers):
        din = spec.conv_out if l == 0 else 2 * hh
        for d in ("fwd", "bwd"):
            t.append((f"lstm{l}.{d}.w_ih", (4 * hh, din), "lstm_wih", hh))
            t.append((f"lstm{l}.{d}.w_hh", (4 * hh, hh), "lstm_whh", hh))
            t.append((f"lstm{l}.{d}.b_ih", (4 * hh,), "lstm_b", hh))
            t.append((f"lstm{l}.{d}.b_hh", (4 * hh,), "lstm_b", hh))
    t.append(("head.weight", (spec.num_classes, 2 * hh), "head_w", 2 * hh))
    t.append(("head.bias", (spec.num_classes,), "head_b", 2 * hh))
    if spec.embed_num:
        t.append(("embeddings_layer.weight", (spec.embed_num + 1, 2 * spec.conv_out), "style_embed", spec.conv_out))
    return t


def num_weight_floats(spec: NetSpec) -> int:
    return sum(int(np.prod(s)) for _n, s, _k, _f in tensor_table(spec))


# ------------------------------------------------------------------ generator

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, stream: int, n: int) -> np.ndarray:
    """n float64 values in [0,1) (24-bit resolution) for (seed, stream), element-indexed."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))],
                                   dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64) + base
    h = splitmix64(idx)
    return (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def generate_weights(spec: NetSpec, seed: int, head_gain: float = 20.0,
                     lstm_gain: float = 3.0, blank_bias: float = 9.0, sa_gain: float = 1.0,
                     boundary_bias: float = 36.0, embed_gain: float = 0.8, walk_gain: float = 12.0,
                     walk_stride: int = 7) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights (no real pero checkpoint exists offline).
    He-uniform for conv layers so activations keep their scale through the
    ReLU stack; torch-default U(-1/sqrt(H), 1/sqrt(H)) * lstm_gain for the LSTM;
    head scaled so logits span several units (a 1e-3 logit tolerance and the
    p<1e-4 sparsification are then meaningful).

    "vgg_sa_s2s": an autoregressive decoder with purely random weights falls into a fixed point after a
    few steps (it repeats one symbol until the length limit), which would exercise nothing.  The output
    projection therefore gets a deterministic structure on top of a small random part:
        dec.out.weight[c] = 0.1 * random + walk_gain * embed[pred(c)] / (embed_gain * sqrt(E)),
    pred(c) = (c - walk_stride) mod (C - 1), so that the most likely next symbol is "previous symbol +
    walk_stride" and the encoder / attention / feed-forward contributions decide where a line leaves that
    walk; boundary_bias sets how often such a departure ends the line.  The result is lines that end after
    a few symbols, lines that run into the reference's length limit, and top-2 margins between 1e-2 and 5."""
    out: Dict[str, np.ndarray] = {}
    if spec.arch in (ARCH_SA, ARCH_S2S):
        head_gain = head_gain * 0.4        # the encoder output is LayerNorm'ed (unit scale), the LSTM's is in (-1, 1)
        blank_bias = blank_bias * 0.6
    for ti, (name, shape, kind, fan_in) in enumerate(tensor_table(spec)):
        n = int(np.prod(shape))
        u = uniform01(seed, ti + 1, n)
        if kind == "conv_w":
            a = (6.0 / fan_in) ** 0.5
            v = (2.0 * u - 1.0) * a
        elif kind == "bias":
            v = (2.0 * u - 1.0) * 0.05
        elif kind == "bn_gamma":
            v = 0.8 + 0.4 * u
        elif kind == "bn_beta":
            v = (2.0 * u - 1.0) * 0.1
        elif kind == "bn_mean":
            v = 0.25 + 0.2 * u          # ~ mean of the LeakyReLU output it normalises
        elif kind == "bn_var":
            v = 0.05 + 0.1 * u          # small running_var -> BN amplifies the input-dependent part
        elif kind == "lstm_wih":
            v = (2.0 * u - 1.0) * (lstm_gain / fan_in ** 0.5)
        elif kind in ("lstm_whh", "lstm_b"):
            v = (2.0 * u - 1.0) * (1.0 / fan_in ** 0.5)   # torch default; keeps the recurrence contractive
        elif kind == "sa_w":
            v = (2.0 * u - 1.0) * (sa_gain * (3.0 / fan_in) ** 0.5)      # variance sa_gain^2 / fan_in
        elif kind == "sa_b":
            v = (2.0 * u - 1.0) * 0.05
        elif kind == "ln_w":
            v = 0.8 + 0.4 * u
        elif kind == "ln_b":
            v = (2.0 * u - 1.0) * 0.1
        elif kind == "head_w":
            v = (2.0 * u - 1.0) * (head_gain / fan_in ** 0.5)
        elif kind == "head_b":
            v = (2.0 * u - 1.0) * 0.5
            v[-1] = blank_bias          # CTC nets emit blank on most frames
        elif kind == "embed":
            v = (2.0 * u - 1.0) * (embed_gain * 3.0 ** 0.5)      # variance embed_gain^2 (nn.Embedding: N(0, 1))
        elif kind == "style_embed":
            v = (2.0 * u - 1.0) * 0.6       # scale 1 + s in (0.4, 1.6), shift in (-0.6, 0.6): every row changes the text
        elif kind == "s2s_b":
            v = (2.0 * u - 1.0) * 0.5
            v[-2] = boundary_bias       # sentence boundary (C-2): random nets must end their lines at some point
            v[-1] = -30.0               # the "ignore" class (C-1) is never predicted by a trained model
        else:  # pragma: no cover
            raise AssertionError(kind)
        out[name] = v.astype(np.float32).reshape(shape)
    if spec.arch == ARCH_S2S:
        c, e = spec.num_classes, spec.conv_out
        pred = (np.arange(c - 1) - walk_stride) % (c - 1)
        w = out["dec.out.weight"].astype(np.float64) * 0.1
        w[:c - 1] += walk_gain * out["dec.embed.weight"][pred].astype(np.float64) / (embed_gain * e ** 0.5)
        out["dec.out.weight"] = w.astype(np.float32)
    return out


# ------------------------------------------------------------------ blob I/O

def pack_weights(spec: NetSpec, weights: Dict[str, np.ndarray]) -> np.ndarray:
    """Flatten to one float32 vector in tensor_table order (the pocr_create() layout)."""
    parts = []
    for name, shape, _k, _f in tensor_table(spec):
        w = np.ascontiguousarray(weights[name], dtype=np.float32)
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {w.shape}")
        parts.append(w.reshape(-1))
    return np.concatenate(parts)


def unpack_weights(spec: NetSpec, flat: np.ndarray) -> Dict[str, np.ndarray]:
    flat = np.asarray(flat, dtype=np.float32).reshape(-1)
    if flat.size != num_weight_floats(spec):
        raise ValueError(f"weight blob has {flat.size} floats, spec needs {num_weight_floats(spec)}")
    out, off = {}, 0
    for name, shape, _k, _f in tensor_table(spec):
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    return out


def save_blob(path: str, spec: NetSpec, weights: Dict[str, np.ndarray]) -> None:
    """File = MAGIC | u32 header_len | JSON spec | float32 data (tensor_table order)."""
    hdr = json.dumps(spec.to_json(), sort_keys=True).encode("utf8")
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(hdr)))
        f.write(hdr)
        f.write(pack_weights(spec, weights).tobytes())


def load_blob(path: str) -> Tuple[NetSpec, Dict[str, np.ndarray]]:
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a POCRW001 weight blob")
        (n,) = struct.unpack("<I", f.read(4))
        spec = NetSpec.from_json(json.loads(f.read(n).decode("utf8")))
        flat = np.frombuffer(f.read(), dtype=np.float32)
    return spec, unpack_weights(spec, flat)
