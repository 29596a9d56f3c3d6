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

from pero_ocr_amd.netspec import BN_EPS, CONV_PLAN, LEAKY_SLOPE, NetSpec


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

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        f = self.features(x)
        y, _ = self.lstm(f.permute(0, 2, 1))
        return self.head(y).permute(0, 2, 1)


def load_lstm_weights(lstm: nn.LSTM, spec: NetSpec, weights: Dict[str, np.ndarray]) -> None:
    for l in range(spec.lstm_layers):
        for d, suffix in (("fwd", ""), ("bwd", "_reverse")):
            for ours, theirs in (("w_ih", "weight_ih"), ("w_hh", "weight_hh"),
                                 ("b_ih", "bias_ih"), ("b_hh", "bias_hh")):
                getattr(lstm, f"{theirs}_l{l}{suffix}").data = torch.from_numpy(
                    weights[f"lstm{l}.{d}.{ours}"].copy())


def forward_logits(net: OracleNet, batch_u8_nhwc: np.ndarray) -> np.ndarray:
    """u8 [n,H,W,3] -> f32 [n,C,T], normalised exactly as pytorch_ocr_engine.py:61-62."""
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(batch_u8_nhwc)).float() / 255.0
        return net(x.permute(0, 3, 1, 2)).numpy()
