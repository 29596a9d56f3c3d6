#!/usr/bin/env python3
"""Convert a PyTorch model with the "vgg_blstm_ctc" topology (pero_ocr_amd/netspec.py) into
a POCRW001 weight blob for the MI355X engine.

Works on any nn.Module / TorchScript module whose parameters appear in this order:
9 x Conv2d(3x3) [+ one BatchNorm2d(512)], the (H/8)x1 aggregation Conv2d, an nn.LSTM
(bidirectional, batch_first irrelevant), a final Linear (or 1x1 Conv1d/Conv2d) to C classes.
Usage: python tools/export_weights.py model.pt out.pocrw --height 40

With --transformer the input is a state_dict of the reference's TransformerOCR
(pero_ocr/ocr_engine/transformer.py:487-508, what TransformerEngineLineOCR loads with
net.load_state_dict(torch.load(checkpoint)), transformer_ocr_engine.py:28) and the output is a
"vgg_sa_s2s" blob for pero_ocr_amd's TransformerEngineLineOCR.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd import netspec  # noqa: E402


def state_to_weights(state: dict, height: int = 40):
    """state: name -> array (a state_dict).  Returns (spec, weights)."""
    items = [(k, np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v)) for k, v in state.items()]
    conv_w = [(k, v) for k, v in items if v.ndim == 4 and v.shape[2:] == (3, 3)]
    if len(conv_w) != 9:
        raise ValueError(f"expected 9 3x3 conv weights, found {len(conv_w)}")
    w = {}

    def bias_of(wname):
        b = wname.rsplit("weight", 1)[0] + "bias"
        return dict(items)[b]
    for i, (k, v) in enumerate(conv_w, start=1):
        cin, cout = netspec.CONV_PLAN[i - 1][:2]
        if v.shape[:2] != (cout, cin):
            raise ValueError(f"{k}: expected [{cout},{cin},3,3], got {v.shape}")
        w[f"conv{i}.weight"], w[f"conv{i}.bias"] = v, bias_of(k)
    d = dict(items)
    bn_mean = [k for k in d if k.endswith("running_mean")]
    if len(bn_mean) != 1:
        raise ValueError("expected exactly one BatchNorm2d")
    p = bn_mean[0][:-len("running_mean")]
    w["bn.gamma"], w["bn.beta"], w["bn.mean"], w["bn.var"] = d[p + "weight"], d[p + "bias"], d[p + "running_mean"], d[p + "running_var"]
    agg = [(k, v) for k, v in items if v.ndim == 4 and v.shape[2:] == (height // 8, 1)]
    if len(agg) != 1:
        raise ValueError("expected exactly one aggregation conv")
    w["agg.weight"], w["agg.bias"] = agg[0][1], bias_of(agg[0][0])
    conv_out = agg[0][1].shape[0]
    layers = sorted({int(k.split("_l")[1].split("_")[0]) for k in d if "weight_ih_l" in k})
    hidden = None
    for l in layers:
        for ours, sfx in (("fwd", ""), ("bwd", "_reverse")):
            pre = [k for k in d if k.endswith(f"weight_ih_l{l}{sfx}")][0][:-len(f"weight_ih_l{l}{sfx}")]
            for a, b in (("w_ih", "weight_ih"), ("w_hh", "weight_hh"), ("b_ih", "bias_ih"), ("b_hh", "bias_hh")):
                w[f"lstm{l}.{ours}.{a}"] = d[f"{pre}{b}_l{l}{sfx}"]
            hidden = w[f"lstm{l}.{ours}.w_hh"].shape[1]
    head = [(k, v) for k, v in items if k.endswith("weight") and v.reshape(v.shape[0], -1).shape[1] == 2 * hidden
            and v.ndim in (2, 3, 4) and not k.split(".")[-1].startswith("weight_")]
    if not head:
        raise ValueError("no output projection found")
    hk, hv = head[-1]
    w["head.weight"], w["head.bias"] = hv.reshape(hv.shape[0], -1), bias_of(hk)
    spec = netspec.NetSpec(num_classes=int(hv.shape[0]), height=height, conv_out=int(conv_out),
                           lstm_hidden=int(hidden), lstm_layers=len(layers))
    return spec, {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def transformer_state_to_weights(state: dict, height: int = 40, heads: int = 8):
    """state_dict of the reference's TransformerOCR -> (spec, weights) of arch "vgg_sa_s2s".
    Parameter names are the reference's attribute paths: encoder_frontend.* (conv stack + aggregation conv),
    encoder.input_norm / encoder.trans_encoder.layers.N.*, trans_decoder.layers.N.{self_attn,multihead_attn,
    linear1,linear2,norm1,norm2,norm3}.*, dec_embeder.weight, dec_out_proj.*."""
    d = {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in state.items()}
    items = list(d.items())
    w = {}
    conv_w = [(k, v) for k, v in items if v.ndim == 4 and v.shape[2:] == (3, 3)]
    if len(conv_w) != 9:
        raise ValueError(f"expected 9 3x3 conv weights, found {len(conv_w)}")
    for i, (k, v) in enumerate(conv_w, start=1):
        cin, cout = netspec.CONV_PLAN[i - 1][:2]
        if v.shape[:2] != (cout, cin):
            raise ValueError(f"{k}: expected [{cout},{cin},3,3], got {v.shape} (only conv_subsampling [8, 4] is built)")
        w[f"conv{i}.weight"], w[f"conv{i}.bias"] = v, d[k[:-len("weight")] + "bias"]
    bn = [k for k in d if k.endswith("running_mean")]
    if len(bn) != 1:
        raise ValueError("expected exactly one BatchNorm2d")
    p = bn[0][:-len("running_mean")]
    w["bn.gamma"], w["bn.beta"], w["bn.mean"], w["bn.var"] = d[p + "weight"], d[p + "bias"], d[p + "running_mean"], d[p + "running_var"]
    agg = [(k, v) for k, v in items if v.ndim == 4 and v.shape[2:] == (height // 8, 1)]
    if len(agg) != 1:
        raise ValueError("expected exactly one aggregation conv")
    w["agg.weight"], w["agg.bias"] = agg[0][1], d[agg[0][0][:-len("weight")] + "bias"]
    e = int(agg[0][1].shape[0])
    w["sa.norm.weight"], w["sa.norm.bias"] = d["encoder.input_norm.weight"], d["encoder.input_norm.bias"]

    def count(prefix):
        return len({k[len(prefix):].split(".")[0] for k in d if k.startswith(prefix)})
    n_enc, n_dec = count("encoder.trans_encoder.layers."), count("trans_decoder.layers.")
    for l in range(n_enc):
        p = f"encoder.trans_encoder.layers.{l}."
        w[f"sa{l}.in_proj.weight"], w[f"sa{l}.in_proj.bias"] = d[p + "self_attn.in_proj_weight"], d[p + "self_attn.in_proj_bias"]
        w[f"sa{l}.out_proj.weight"], w[f"sa{l}.out_proj.bias"] = d[p + "self_attn.out_proj.weight"], d[p + "self_attn.out_proj.bias"]
        for ours, theirs in (("lin1", "linear1"), ("lin2", "linear2"), ("norm1", "norm1"), ("norm2", "norm2")):
            w[f"sa{l}.{ours}.weight"], w[f"sa{l}.{ours}.bias"] = d[p + theirs + ".weight"], d[p + theirs + ".bias"]
    for l in range(n_dec):
        p = f"trans_decoder.layers.{l}."
        for ours, theirs in (("self", "self_attn"), ("cross", "multihead_attn")):
            w[f"dec{l}.{ours}.in_proj.weight"], w[f"dec{l}.{ours}.in_proj.bias"] = d[p + theirs + ".in_proj_weight"], d[p + theirs + ".in_proj_bias"]
            w[f"dec{l}.{ours}.out_proj.weight"], w[f"dec{l}.{ours}.out_proj.bias"] = d[p + theirs + ".out_proj.weight"], d[p + theirs + ".out_proj.bias"]
        for ours, theirs in (("lin1", "linear1"), ("lin2", "linear2"), ("norm1", "norm1"), ("norm2", "norm2"), ("norm3", "norm3")):
            w[f"dec{l}.{ours}.weight"], w[f"dec{l}.{ours}.bias"] = d[p + theirs + ".weight"], d[p + theirs + ".bias"]
    w["dec.embed.weight"] = d["dec_embeder.weight"]
    w["dec.out.weight"], w["dec.out.bias"] = d["dec_out_proj.weight"], d["dec_out_proj.bias"]
    spec = netspec.NetSpec(num_classes=int(w["dec.out.weight"].shape[0]), height=height, conv_out=e, arch=netspec.ARCH_S2S,
                           sa_layers=n_enc, sa_heads=heads, sa_ff=int(w["sa0.lin1.weight"].shape[0]), dec_layers=n_dec)
    return spec, {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("out")
    ap.add_argument("--height", type=int, default=40)
    ap.add_argument("--transformer", action="store_true", help="input is a TransformerOCR state_dict (seq2seq engine)")
    ap.add_argument("--heads", type=int, default=8, help="attention heads (not recoverable from a state_dict)")
    a = ap.parse_args()
    import torch
    try:
        m = torch.jit.load(a.model, map_location="cpu")
    except Exception:
        m = torch.load(a.model, map_location="cpu", weights_only=False)
    state = m if isinstance(m, dict) else m.state_dict()
    spec, weights = (transformer_state_to_weights(state, a.height, a.heads) if a.transformer
                     else state_to_weights(state, a.height))
    netspec.save_blob(a.out, spec, weights)
    print(f"wrote {a.out}: {spec}")


if __name__ == "__main__":
    main()
