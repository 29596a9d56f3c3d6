"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path
(pero_ocr_amd/*); only tests/ and oracle/gen_golden_s2s.py use it.

PyTorch-CPU fp32 restatement of the reference's transformer (sequence-to-sequence) line
recogniser, SURVEY.md section 8 row f-3:

  * network   TransformerOCR.encode + the cached greedy decoding loop,
              pero_ocr/ocr_engine/transformer.py:388-463 (DecoderLayer.infer), :155-313
              (CustomMultiheadAttention.cached_forward), :466-508, and
              pero_ocr/ocr_engine/transformer_ocr_engine.py:32-89 (run_ocr / transcribe_batch);
  * host      the "transformer" branches of BaseEngineLineOCR.process_lines,
              pero_ocr/ocr_engine/line_ocr_engine.py:84-85,95-119,131-142,161-162 and
              merge_transcriptions_and_logits / find_best_overlap :180-211
              (Levenshtein distance: pero_ocr/sequence_alignment.py:4-13).

Written batch-first ([n, ...]; the reference is sequence-first) and per line where the
reference is per batch - lines never interact inside the network.

Parity status: pinned against the reference's own classes (TransformerEngineLineOCR driving
transformer.build_net's TransformerOCR, CPU) by oracle/gen_golden_s2s.py; fixtures
tests/golden/s2s_*.{json,npz}.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from pero_ocr_amd.netspec import LN_EPS, NetSpec
from oracle.model_oracle import OracleNet

MIN_DECODER_WIDTH = 1088        # transformer_ocr_engine.py:36-40: narrower batches are centred in 1088 columns


def sinusoid(rows: int, e: int) -> torch.Tensor:
    pos = torch.arange(0, rows, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, e, 2).float() * (-math.log(10000.0) / e))
    pe = torch.zeros(rows, e)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class OracleS2S:
    def __init__(self, spec: NetSpec, weights: Dict[str, np.ndarray]):
        self.spec = spec
        self.enc = OracleNet(spec, weights)
        self.w = {k: torch.from_numpy(v.copy()) for k, v in weights.items() if k.startswith("dec")}
        self.boundary = spec.num_classes - 2           # transformer_ocr_engine.py:18
        self.ignore = spec.num_classes - 1             # :19

    # ---- encoder: conv frontend + LineSelfAttentionEncoder -> memory [n, T, E]
    def encode(self, batch_u8_nhwc: np.ndarray) -> torch.Tensor:
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(batch_u8_nhwc)).float()
            x /= 255.0                                               # transformer_ocr_engine.py:51-52
            return self.enc.encoder_stages(self.enc.features(x.permute(0, 3, 1, 2)))[-1]

    def _attend(self, q, k, v):
        """q [n,E] (already projected), k/v [n,S,E] -> [n,E]; per head softmax((q*d^-0.5) k^T) v (transformer.py:270-285)"""
        n, e = q.shape
        h = self.spec.sa_heads
        d = e // h
        q = (q * (float(d) ** -0.5)).view(n, h, 1, d)
        k = k.view(n, -1, h, d).permute(0, 2, 1, 3)
        v = v.view(n, -1, h, d).permute(0, 2, 1, 3)
        p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
        return (p @ v).reshape(n, e)

    def decode(self, memory: torch.Tensor, width_px: int):
        """Greedy cached decoding of one batch (transformer_ocr_engine.py:49-89).
        -> (tokens [steps_kept, n] as fed back, logits [n, steps, C], details)"""
        sp, w = self.spec, self.w
        e = sp.conv_out
        n = memory.shape[0]
        pe = sinusoid(width_px // 4 + 8, e)
        ln = lambda x, name: F.layer_norm(x, (e,), w[name + ".weight"], w[name + ".bias"], LN_EPS)
        with torch.no_grad():
            mem_kv = []
            for l in range(sp.dec_layers):                       # cached once per batch (transformer.py:237-247)
                wi, bi = w[f"dec{l}.cross.in_proj.weight"], w[f"dec{l}.cross.in_proj.bias"]
                kv = F.linear(memory, wi[e:], bi[e:])
                mem_kv.append((kv[..., :e], kv[..., e:]))
            self_k = [[] for _ in range(sp.dec_layers)]
            self_v = [[] for _ in range(sp.dec_layers)]
            prev = torch.full((n,), self.boundary, dtype=torch.long)
            alive = torch.ones(n, dtype=torch.bool)
            fed, logits = [prev], []
            while True:
                s = len(logits)
                x = w["dec.embed.weight"][prev] + pe[s]
                for l in range(sp.dec_layers):
                    g = lambda k: w[f"dec{l}.{k}"]
                    qkv = F.linear(x, g("self.in_proj.weight"), g("self.in_proj.bias"))
                    self_k[l].append(qkv[:, e:2 * e])
                    self_v[l].append(qkv[:, 2 * e:])
                    a = self._attend(qkv[:, :e], torch.stack(self_k[l], 1), torch.stack(self_v[l], 1))
                    x = ln(x + F.linear(a, g("self.out_proj.weight"), g("self.out_proj.bias")), f"dec{l}.norm1")
                    q = F.linear(x, g("cross.in_proj.weight")[:e], g("cross.in_proj.bias")[:e])
                    a = self._attend(q, *mem_kv[l])
                    x = ln(x + F.linear(a, g("cross.out_proj.weight"), g("cross.out_proj.bias")), f"dec{l}.norm2")
                    ff = F.linear(F.relu(F.linear(x, g("lin1.weight"), g("lin1.bias"))), g("lin2.weight"), g("lin2.bias"))
                    x = ln(x + ff, f"dec{l}.norm3")
                step_logits = F.linear(x, w["dec.out.weight"], w["dec.out.bias"])
                logits.append(step_logits)
                sample = torch.argmax(step_logits, dim=-1)
                alive &= sample != self.boundary
                if not bool(alive.any()):
                    break
                if len(fed) > width_px // 4:                       # :77 "four pixels per letter is already ridiculous"
                    break
                fed.append(sample)
                prev = sample
        return torch.stack(fed[1:]) if len(fed) > 1 else torch.zeros((0, n), dtype=torch.long), torch.stack(logits, 1)

    def labels_of(self, tokens: torch.Tensor) -> List[List[int]]:
        """postprocess_decoded (transformer_ocr_engine.py:91-105): cut at the first boundary, skip 'ignore'."""
        out = []
        for i in range(tokens.shape[1]):
            row = []
            for s in tokens[:, i].tolist():
                if s == self.boundary:
                    break
                if s != self.ignore:
                    row.append(s)
            out.append(row)
        return out

    def run_ocr(self, batch_u8_nhwc: np.ndarray, characters: Sequence[str]):
        """-> (strings, logits [n, steps, C]) for one padded batch, as TransformerEngineLineOCR.run_ocr"""
        b = np.asarray(batch_u8_nhwc)
        if b.shape[2] < MIN_DECODER_WIDTH:
            wide = np.zeros((b.shape[0], b.shape[1], MIN_DECODER_WIDTH, 3), dtype=b.dtype)
            s = (MIN_DECODER_WIDTH - b.shape[2]) // 2
            wide[:, :, s:s + b.shape[2]] = b
            b = wide
        tokens, logits = self.decode(self.encode(b), b.shape[2])
        return ["".join(characters[c] for c in row) for row in self.labels_of(tokens)], logits.numpy()


# ------------------------------------------------------------------------------------------ host logic

def edit_distance(a: Sequence, b: Sequence) -> int:
    """Unit-cost Levenshtein distance (what pero_ocr/sequence_alignment.py:4-13 computes)."""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, start=1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, start=1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[-1]


def best_overlap(left: str, right: str) -> int:
    """find_best_overlap (line_ocr_engine.py:196-211): the overlap length i in 1..min(len) whose
    suffix/prefix pair has the lowest character error rate; ties keep the shortest; 0 if none is below 1."""
    best_cer, best = 1, 0
    for i in range(1, min(len(left), len(right)) + 1):
        cer = edit_distance(left[-i:], right[:i]) / i
        if cer < best_cer:
            best_cer, best = cer, i
    return best


def merge_parts(texts: Sequence[str], logits: Sequence[np.ndarray]) -> Tuple[str, np.ndarray]:
    """merge_transcriptions_and_logits (line_ocr_engine.py:180-193).  Note the reference's slice
    `[:-overlap // 2]` parses as [: (-overlap) // 2] = [: -ceil(overlap / 2)], which for overlap 0 is [:0]."""
    text = texts[0]
    lg = logits[0][:len(texts[0])]
    for t, l in zip(texts[1:], logits[1:]):
        l = l[:len(t)]
        ov = best_overlap(text, t)
        cut = (-ov) // 2
        text = text[:cut] + t[ov // 2:]
        lg = np.concatenate([lg[:cut], l[ov // 2:]], axis=0)
    return text, lg


def split_line(width: int, max_line_width: int) -> List[Tuple[int, int]]:
    """Column spans of the parts of an over-long line (line_ocr_engine.py:96-113)."""
    if width <= max_line_width:
        return [(0, width)]
    step = max_line_width - max_line_width // 4
    spans, start, end = [], 0, max_line_width
    while end < width:
        spans.append((start, end))
        start += step
        end += step
    spans.append((start, min(end, width)))
    return spans


def plan_batches(widths: Sequence[int], max_px: int, max_line_width, pad: int = 32):
    """[(line ids, max_width)] in processing order (line_ocr_engine.py:79-90 with the transformer clamp :84-85)."""
    order = sorted(range(len(widths)), key=lambda i: -int(widths[i]))
    out, pos = [], 0
    while pos < len(order):
        mw = -(-int(widths[order[pos]]) // 32) * 32
        mw = min(mw, max_line_width + 2 * pad)
        take = max(1, int(max_px) // int(mw))
        out.append((order[pos:pos + take], int(mw)))
        pos += take
    return out


def process_lines(model: OracleS2S, lines: Sequence[np.ndarray], characters: Sequence[str], height: int, max_px: int,
                  max_line_width, pad: int = 32):
    """Dense-logit form of process_lines for model_type == "transformer".
    -> (texts, logits [len_i.., C] per line (all decoding steps of the batch), coords, extras)"""
    n = len(lines)
    texts, logits, coords = [None] * n, [None] * n, [None] * n
    extras = {"plan": [], "steps": []}
    for ids, mw in plan_batches([l.shape[1] for l in lines], max_px, max_line_width, pad):
        images, spans = [], []
        for i in ids:
            parts = split_line(lines[i].shape[1], max_line_width)
            images += [lines[i][:, a:b] for a, b in parts]
            spans.append(len(parts))
        batch = np.zeros((len(images), height, mw + 2 * pad, 3), dtype=np.uint8)
        for row, img in zip(batch, images):
            row[:, pad:pad + img.shape[1]] = img
        batch = batch[:, :, :max_px]
        out_t, out_l = model.run_ocr(batch, characters)
        extras["plan"].append((list(ids), mw, spans))
        extras["steps"].append(out_l.shape[1])
        k = 0
        for i, span in zip(ids, spans):
            texts[i], logits[i] = merge_parts(out_t[k:k + span], out_l[k:k + span])
            coords[i] = [0, len(texts[i])]
            k += span
    return texts, logits, coords, extras
