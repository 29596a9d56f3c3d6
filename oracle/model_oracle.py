"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path
(pero_ocr_amd/*); only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may use it.

PyTorch-CPU fp32 restatement of the line recogniser network the reference drives
as an opaque TorchScript module (pero_ocr/ocr_engine/pytorch_ocr_engine.py:56,64-69;
contract "[N,3,H,W] f32 -> [N,C,T] f32, blank last", :12).  Topology:

  * conv backbone  = pero_ocr/ocr_engine/transformer.py: create_vgg_block_2d :51-72,
    VGG_conv_module :75-148 built with subsampling=(8,4), layers_2d=17 (first 17
    entries of torchvision VGG16 .features), base_channels=64, conv_blocks=4, i.e.
    7 x (conv3x3 + ReLU) with pools (2,2),(2,2),(2,1), then conv256->512 + LeakyReLU,
    conv512->512 + LeakyReLU, identity pool, BatchNorm2d(512);
  * aggregation conv (H/8 x 1) + LeakyReLU = ConvolutionalEncoder :335-363;
  * BiLSTM stack and linear head are NOT in the reference tree: torch.nn.LSTM /
    torch.nn.Linear semantics.

Parity status: the conv/aggregation topology is pinned against the reference's
own modules by oracle/gen_golden.py (which instantiates transformer.ConvolutionalEncoder
and runs it through the reference PytorchEngineLineOCR); the fixtures it wrote are
in tests/golden/.  The arithmetic underneath is stock PyTorch (oneDNN) fp32.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
from torch import nn

from pero_ocr_amd.netspec import ARCH_S2S, ARCH_SA, BN_EPS, CONV_PLAN, LEAKY_SLOPE, LN_EPS, NetSpec


class OracleNet(nn.Module):
    def __init__(self, spec: NetSpec, weights: Dict[str, np.ndarray]):
        super().__init__()
        self.spec = spec
        layers = []
        for i, (cin, cout, act, pool) in enumerate(CONV_PLAN, start=1):
            conv = nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1)
            conv.weight.data = torch.from_numpy(weights[f"conv{i}.weight"].copy())
            conv.bias.data = torch.from_numpy(weights[f"conv{i}.bias"].copy())
            layers.append(conv)
            layers.append(nn.ReLU() if act == "relu" else nn.LeakyReLU(LEAKY_SLOPE))
            if pool != (1, 1):
                layers.append(nn.MaxPool2d(kernel_size=pool, stride=pool))
        c_last = CONV_PLAN[-1][1]
        bn = nn.BatchNorm2d(c_last, eps=BN_EPS)
        bn.weight.data = torch.from_numpy(weights["bn.gamma"].copy())
        bn.bias.data = torch.from_numpy(weights["bn.beta"].copy())
        bn.running_mean.data = torch.from_numpy(weights["bn.mean"].copy())
        bn.running_var.data = torch.from_numpy(weights["bn.var"].copy())
        layers.append(bn)
        self.backbone = nn.Sequential(*layers)
        self.agg = nn.Conv2d(c_last, spec.conv_out, kernel_size=(spec.agg_height, 1), stride=1, padding=0)
        self.agg.weight.data = torch.from_numpy(weights["agg.weight"].copy())
        self.agg.bias.data = torch.from_numpy(weights["agg.bias"].copy())
        self.agg_act = nn.LeakyReLU(LEAKY_SLOPE)
        self.embeddings_layer = None
        if spec.embed_num:                   # pytorch_ocr_engine.py:49-50 reads model.embeddings_layer.weight.shape[0]
            self.embeddings_layer = nn.Embedding(spec.embed_num + 1, 2 * spec.conv_out)
            self.embeddings_layer.weight.data = torch.from_numpy(weights["embeddings_layer.weight"].copy())
        self.sa = None
        if spec.arch == ARCH_S2S:            # encoder only; the decoder lives in oracle/s2s_oracle.py
            self.sa = {k: torch.from_numpy(v.copy()) for k, v in weights.items() if k.startswith("sa")}
            self.eval()
            return
        if spec.arch == ARCH_SA:
            self.sa = {k: torch.from_numpy(v.copy()) for k, v in weights.items() if k.startswith("sa")}
            self.head = nn.Linear(spec.conv_out, spec.num_classes)
        else:
            self.lstm = nn.LSTM(spec.conv_out, spec.lstm_hidden, num_layers=spec.lstm_layers,
                                bidirectional=True, batch_first=True)
            load_lstm_weights(self.lstm, spec, weights)
            self.head = nn.Linear(2 * spec.lstm_hidden, spec.num_classes)
        self.head.weight.data = torch.from_numpy(weights["head.weight"].copy())
        self.head.bias.data = torch.from_numpy(weights["head.bias"].copy())
        self.eval()

    def features(self, x: torch.Tensor) -> torch.Tensor:
        """[N,3,H,W] -> [N,E,T]"""
        f = self.agg_act(self.agg(self.backbone(x)))
        return f.squeeze(2)

    def encoder_stages(self, f: torch.Tensor):
        """Self-attention encoder restated from the documented equations of
        LineSelfAttentionEncoder (pero_ocr/ocr_engine/transformer.py:366-385): LayerNorm(E, 1e-5),
        + sinusoidal PE (:316-332), then post-norm nn.TransformerEncoderLayer blocks (ReLU FFN,
        dropout 0, no mask): x = LN1(x + MHA(x)); x = LN2(x + W2 relu(W1 x)).  f: [N,E,T].
        Returns the list [after input norm+PE, after layer 0, ...] as [N,T,E] tensors."""
        import math
        import torch.nn.functional as F
        sp, w = self.spec, self.sa
        if f.dtype != torch.float32:                                  # oracle/gen_truth_rows.py runs the restatement in float64
            w = {k: v.to(f.dtype) for k, v in w.items()}
        e, h = sp.conv_out, sp.sa_heads
        d = e // h
        x = f.permute(0, 2, 1)                                        # [N,T,E]; lines are independent
        n, t, _ = x.shape
        position = torch.arange(0, t, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, e, 2).float() * (-math.log(10000.0) / e))
        pe = torch.zeros(t, e)
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        pe = pe.to(f.dtype)                                           # the table itself is a float32 constant of the model
        x = F.layer_norm(x, (e,), w["sa.norm.weight"], w["sa.norm.bias"], LN_EPS) + pe
        outs = [x]
        for l in range(sp.sa_layers):
            g = lambda k: w[f"sa{l}.{k}"]
            qkv = F.linear(x, g("in_proj.weight"), g("in_proj.bias")).view(n, t, 3, h, d)
            q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))           # [N,h,T,d]
            att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v   # [N,h,T,d]
            att = att.permute(0, 2, 1, 3).reshape(n, t, e)
            x = F.layer_norm(x + F.linear(att, g("out_proj.weight"), g("out_proj.bias")), (e,),
                             g("norm1.weight"), g("norm1.bias"), LN_EPS)
            ff = F.linear(F.relu(F.linear(x, g("lin1.weight"), g("lin1.bias"))), g("lin2.weight"), g("lin2.bias"))
            x = F.layer_norm(x + ff, (e,), g("norm2.weight"), g("norm2.bias"), LN_EPS)
            outs.append(x)
        return outs

    def forward(self, x: torch.Tensor, ids=None) -> torch.Tensor:
        f = self.features(x)
        if self.embeddings_layer is not None:                # `model(batch_data, ids_embedding)`, pytorch_ocr_engine.py:64-66
            f = apply_style_embedding(f, self.embeddings_layer(ids), self.spec.conv_out)
        if self.sa is not None:
            return self.head(self.encoder_stages(f)[-1]).permute(0, 2, 1)
        y, _ = self.lstm(f.permute(0, 2, 1))
        return self.head(y).permute(0, 2, 1)


def apply_style_embedding(f: torch.Tensor, emb: torch.Tensor, e: int) -> torch.Tensor:
    """f [N, E, T], emb [N, 2E] -> f * (1 + scale) + shift, per line and channel (netspec.NetSpec.embed_num)."""
    return f * (1.0 + emb[:, :e]).unsqueeze(2) + emb[:, e:].unsqueeze(2)


def load_lstm_weights(lstm: nn.LSTM, spec: NetSpec, weights: Dict[str, np.ndarray]) -> None:
    for l in range(spec.lstm_layers):
        for d, suffix in (("fwd", ""), ("bwd", "_reverse")):
            for ours, theirs in (("w_ih", "weight_ih"), ("w_hh", "weight_hh"),
                                 ("b_ih", "bias_ih"), ("b_hh", "bias_hh")):
                getattr(lstm, f"{theirs}_l{l}{suffix}").data = torch.from_numpy(
                    weights[f"lstm{l}.{d}.{ours}"].copy())


def forward_logits(net: OracleNet, batch_u8_nhwc: np.ndarray, embed_id=None) -> np.ndarray:
    """u8 [n,H,W,3] -> f32 [n,C,T], normalised exactly as pytorch_ocr_engine.py:61-62; embed_id as :64-66."""
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(batch_u8_nhwc)).float() / 255.0
        if embed_id is not None:
            ids = torch.LongTensor([int(embed_id)] * x.shape[0])
            return net(x.permute(0, 3, 1, 2), ids).numpy()
        return net(x.permute(0, 3, 1, 2)).numpy()
