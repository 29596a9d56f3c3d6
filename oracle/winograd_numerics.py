"""ORACLE TOOLING (numerics study, CPU only; nothing in the product or the tests uses it).

Would a Winograd F(2,3) along the width - 4 instead of 6 multiplications per two output columns of a 3x3 convolution, i.e.
two thirds of the matrix-pipe work of conv2..conv9 - stay inside the parity bar?    The network is the
oracle's restatement of the reference's backbone (pero_ocr/ocr_engine/transformer.py:51-72, 335-363); only the arithmetic of
the 3x3 layers is varied, everything else is torch fp32:

  ref32      torch conv2d in float32                                   (= the reference's arithmetic)
  hip_direct operands rounded to f16x2's 22 significand bits, products accumulated exactly (float64), one fp32 rounding per
             output value                                                (a model of the shipped MFMA arithmetic, slightly optimistic)
  hip_wino2d F(2x2, 3x3) on the layers of even height (conv2..conv7: 16 instead of 36 multiplications per 2x2 outputs)
  hip_wino   the same arithmetic behind a width-wise F(2,3): V = B^T d in fp32 (one add per value), U = G g in float64 on the
             host, both rounded to 22 bits, four exact-product accumulations, y = A^T M in fp32

and every variant is judged against the float64 network on the same lines: max / rms logit error, rows above 5e-4, frames
whose arg-max differs where the float64 margin is >= 2e-3.

usage: python oracle/winograd_numerics.py [n_chunks] [fixture]        (default: 10 chunks of c3 incl. the chunk of its worst line)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import Golden  # noqa: E402
from oracle import engine_oracle, model_oracle  # noqa: E402


def q22(x: torch.Tensor) -> torch.Tensor:
    """fp32/64 -> the value f16x2 represents: h = f16(x), l = f16((x - h) * 2^11); h + l / 2^11 (returned in float64)."""
    x = x.double()
    h = x.float().half().double()
    l = ((x - h) * 2048.0).float().half().double()
    return h + l / 2048.0


class Direct22(nn.Module):
    def __init__(self, conv: nn.Conv2d):
        super().__init__()
        self.w, self.b = q22(conv.weight.data), conv.bias.data.double()

    def forward(self, x):
        return F.conv2d(q22(x), self.w, self.b, padding=1).float()


class Wino22(nn.Module):
    """F(2,3) along W: y[2t], y[2t+1] from d[2t-1 .. 2t+2]."""

    def __init__(self, conv: nn.Conv2d, quantise=True):
        super().__init__()
        g = conv.weight.data.double()                               # [Cout, Cin, dy, dx]
        g0, g1, g2 = g[..., 0:1], g[..., 1:2], g[..., 2:3]
        u = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2]        # [Cout, Cin, 3, 1] each, float64
        self.u = [q22(k) if quantise else k.float() for k in u]
        self.b = conv.bias.data.float()
        self.quantise = quantise

    def forward(self, x):
        n, c, h, w = x.shape
        assert w % 2 == 0
        xp = F.pad(x.float(), (1, 2, 1, 1))
        d = [xp[..., k:w + k:2] for k in range(4)]
        v = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]    # fp32, one rounding each
        if self.quantise:
            m = [F.conv2d(q22(v[k]), self.u[k]).float() for k in range(4)]
        else:
            m = [F.conv2d(v[k], self.u[k]) for k in range(4)]
        b = self.b.view(1, -1, 1, 1)
        y0 = (m[0] + m[1]) + m[2] + b
        y1 = (m[1] - m[2]) - m[3] + b
        return torch.stack((y0, y1), dim=-1).reshape(n, -1, h, w)


class Wino22x2(nn.Module):
    """F(2x2, 3x3): 16 instead of 36 multiplications per 2x2 outputs; layers of odd height (conv8 / conv9 at H = 5) use the
    width-wise form."""

    def __init__(self, conv: nn.Conv2d):
        super().__init__()
        self.w1d = Wino22(conv)
        g = conv.weight.data.double()                               # [Cout, Cin, dy, dx]
        def gmat(a0, a1, a2):
            return [a0, (a0 + a1 + a2) / 2, (a0 - a1 + a2) / 2, a2]
        gw = gmat(g[..., 0], g[..., 1], g[..., 2])                  # along dx: 4 x [Cout, Cin, dy]
        self.u = [[q22(k)[..., None, None] for k in gmat(q[..., 0], q[..., 1], q[..., 2])] for q in gw]   # u[nu][xi]: [Cout, Cin, 1, 1]
        self.b = conv.bias.data.float()

    def forward(self, x):
        n, c, h, w = x.shape
        if h % 2:
            return self.w1d(x)
        xp = F.pad(x.float(), (1, 2, 1, 2))
        d = [xp[..., k:w + k:2] for k in range(4)]
        vw = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]                  # [n, c, h + 3, w / 2]
        b = self.b.view(1, -1, 1, 1)
        z = []
        for xi in range(4):
            m = []
            for nu in range(4):
                e = [vw[nu][..., k:h + k:2, :] for k in range(4)]
                v = [e[0] - e[2], e[1] + e[2], e[2] - e[1], e[1] - e[3]][xi]
                m.append(F.conv2d(q22(v), self.u[nu][xi]).float())
            z.append(((m[0] + m[1]) + m[2], (m[1] - m[2]) - m[3]))                 # along W, for this xi
        y = [[(z[0][j] + z[1][j]) + z[2][j] + b, (z[1][j] - z[2][j]) - z[3][j] + b] for j in range(2)]     # y[j along W][i along H]
        rows0 = torch.stack((y[0][0], y[1][0]), dim=-1).reshape(n, -1, h // 2, w)
        rows1 = torch.stack((y[0][1], y[1][1]), dim=-1).reshape(n, -1, h // 2, w)
        return torch.stack((rows0, rows1), dim=3).reshape(n, -1, h, w)


def variant(spec, weights, kind):
    net = model_oracle.OracleNet(spec, weights)
    if kind == "ref32":
        return net
    for i, mod in enumerate(net.backbone):
        if isinstance(mod, nn.Conv2d) and mod.in_channels > 3:      # conv1 (K = 27) stays as it is in every variant
            net.backbone[i] = {"hip_direct": Direct22, "hip_wino": Wino22, "hip_wino2d": Wino22x2, "wino32": lambda c: Wino22(c, quantise=False)}[kind](mod)
    return net


def main():
    n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    g = Golden(sys.argv[2] if len(sys.argv) > 2 else "c3")
    spec, weights, crops = g.spec(), g.weights(), g.crops()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    plan = list(g.plan)
    worst_line = 1530 if g.n > 1530 else 0
    picked = [k for k, (ids, _mw) in enumerate(plan) if worst_line in ids]
    step = max(1, len(plan) // max(1, n_chunks - 1))
    picked += [k for k in range(0, len(plan), step) if k not in picked][:n_chunks - 1]
    net64 = model_oracle.OracleNet(spec, weights).double()
    kinds = ["ref32", "wino32", "hip_direct", "hip_wino", "hip_wino2d"]
    nets = {k: variant(spec, weights, k) for k in kinds}
    err = {k: [] for k in kinds}
    flips = {k: 0 for k in kinds}
    frames = 0
    for k in sorted(picked):
        ids, mw = plan[k]
        batch = engine_oracle.assemble_batch(crops, ids, spec.height, mw, 480 * g.batch_size)
        x8 = torch.from_numpy(np.ascontiguousarray(batch))
        with torch.no_grad():
            truth = net64((x8.double() / 255.0).permute(0, 3, 1, 2)).numpy()          # [n, C, T]
            x32 = (x8.float() / 255.0).permute(0, 3, 1, 2)
            srt = np.sort(truth, axis=1)
            safe = (srt[:, -1] - srt[:, -2]) >= 2e-3
            frames += int(safe.sum())
            for kind in kinds:
                out = nets[kind](x32).double().numpy()
                err[kind].append((out - truth).ravel())
                flips[kind] += int((np.argmax(out, axis=1) != np.argmax(truth, axis=1))[safe].sum())
        print(f"chunk {k}: {len(ids)} lines, W_pad {batch.shape[2]}", flush=True)
    print(f"{len(picked)} chunks of {g.name if hasattr(g, 'name') else 'fixture'}, {frames} frames with a float64 margin >= 2e-3")
    for kind in kinds:
        e = np.concatenate(err[kind])
        print(f"{kind:11s} max |logit - float64| {np.abs(e).max():.3e}  rms {np.sqrt(np.mean(e * e)):.3e}  "
              f"entries above 5e-4: {int((np.abs(e) > 5e-4).sum())}  arg-max flips on decidable frames: {flips[kind]}")


if __name__ == "__main__":
    main()
