"""ORACLE TOOLING — runs ONLY in the build container (needs /root/reference).

Imports the reference's TorchParseNet (pero_ocr/layout_engines/torch_parsenet.py; its only non-standard import is cv2,
replaced by a stub whose `resize` serves the calls the fixture makes: factor 1 = a copy) and drives its get_maps
(:37-58: pad to multiples of 64, `* (1/255.)`, `out_map, _ = net(x)`, crop) with the TorchScript of this build's layout
network (oracle/parsenet_oracle.py) filled with seeded weights.  Stores the maps as fixtures tests/golden/parsenet_*.npz and
checks that the restatement (parsenet_oracle.get_maps) reproduces them.  Also records known answers of the reference's
adaptive down-sampling logic (get_maps_with_optimal_resolution / get_med_height, :60-103) on crafted maps.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REFERENCE = "/root/reference"

from pero_ocr_amd import parsenet_spec as ps, synth  # noqa: E402
from oracle import parsenet_oracle as po  # noqa: E402

WEIGHT_SEED = 20261001


def import_reference():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_AREA = 3

    def resize(img, dsize, fx=None, fy=None, interpolation=None):
        assert dsize == (0, 0) and interpolation == cv2.INTER_AREA
        if fx == 1 and fy == 1:
            return img.copy()
        raise NotImplementedError("the fixtures feed pre-resized pages (cv2 is not installed)")

    cv2.resize = resize
    sys.modules["cv2"] = cv2
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    from pero_ocr.layout_engines import torch_parsenet
    return torch_parsenet


def adaptive_cases(ref_cls_factory):
    """Known answers of get_maps_with_optimal_resolution: the network call is replaced by crafted maps."""
    cases = []
    scen = [  # (image shape, init downsample, max_mp, height written into channel 0, fraction of line pixels)
        ((3000, 4000), 4, 5, 12.0, 0.02), ((3000, 4000), 4, 5, 30.0, 0.02), ((3000, 4000), 4, 5, 5.0, 0.02),
        ((3000, 4000), 4, 5, 14.9, 0.02), ((3000, 4000), 4, 5, 20.0, 0.0), ((6000, 8000), 2, 5, 8.0, 0.02),
        ((800, 600), 4, 5, 100.0, 0.05), ((3000, 4000), 4, None, 16.0, 0.02), ((1000, 1500), 1, 5, 3.0, 0.02),
    ]
    for shape, ds, mp, hval, frac in scen:
        eng = ref_cls_factory(ds, mp)
        calls = []

        def fake_get_maps(img, downsample, _calls=calls, _hval=hval, _frac=frac):
            _calls.append(float(downsample))
            h, w = int(img.shape[0] / downsample), int(img.shape[1] / downsample)
            m = np.zeros((h, w, 5), np.float32)
            n = int(h * w * _frac)
            m.reshape(-1, 5)[:n, 2] = 0.9
            m.reshape(-1, 5)[:n, 0] = _hval * (4.0 / downsample) if _frac else 0.0
            return m
        eng.get_maps = fake_get_maps
        img = np.zeros(shape + (3,), np.uint8)
        out, net_ds = eng.get_maps_with_optimal_resolution(img)
        cases.append({"shape": list(shape), "downsample": ds, "max_mp": mp, "height": hval, "frac": frac,
                      "calls": calls, "net_downsample": float(net_ds), "last_downsample": float(eng.last_downsample),
                      "out_shape": list(out.shape)})
    return cases


def main():
    tp = import_reference()
    weights = ps.generate_weights(WEIGHT_SEED)
    torch.set_num_threads(os.cpu_count() or 1)
    # head bias calibrated on data (like the recogniser fixtures): with purely random weights the head's pre-activations
    # sit far from 0 and the probability channels saturate; centre them on a calibration page and store the result
    cal = synth.make_page(5, 256, 384)
    with torch.no_grad():
        x = torch.from_numpy(cal[None]).float().permute(0, 3, 1, 2) * (1 / 255.)
        z = po.ParseNetOracle(weights)(x)[1]
    shift = z.double().mean(dim=(0, 2, 3)).numpy() - np.array([1.0, 0.5, -1.5, -2.0, -1.0])   # lines are the minority class
    weights["head.bias"] = (weights["head.bias"].astype(np.float64) - shift).astype(np.float32)
    net = po.ParseNetOracle(weights)
    with tempfile.TemporaryDirectory() as td:
        torch.jit.script(net).save(os.path.join(td, "parsenet.pt.cpu"))       # CPU path appends ".cpu" (torch_parsenet.py:11-12)
        ref = tp.TorchParseNet(os.path.join(td, "parsenet.pt"), torch.device("cpu"), downsample=1, adaptive_downsample=False)
        meta = {"arch": ps.ARCH, "weight_seed": WEIGHT_SEED, "pages": {}, "torch": torch.__version__}
        arrays = {"override_head.bias": weights["head.bias"]}
        for name, seed, h, w in (("small", 11, 200, 300), ("odd", 12, 65, 129), ("page", 13, 768, 1024)):
            page = synth.make_page(seed, h, w)
            out = np.ascontiguousarray(ref.get_maps(page, 1))                  # the reference's own code path
            mine = po.get_maps(net, page)
            err = float(np.max(np.abs(out - mine)))
            print(f"[{name}] {h}x{w}: reference get_maps vs restatement max |d| = {err:.3e}; "
                  f"channel means {[round(float(out[..., c].mean()), 4) for c in range(5)]}")
            assert out.shape == (h, w, 5) and out.dtype == np.float32 and err < 1e-5
            meta["pages"][name] = {"seed": seed, "height": h, "width": w, "oracle_vs_reference_max_abs": err,
                                   "channel_mean": [float(out[..., c].mean()) for c in range(5)],
                                   "channel_max": [float(out[..., c].max()) for c in range(5)]}
            if h * w <= 100000:
                arrays[f"{name}_maps"] = out
            else:                               # big page: every 4th pixel + full-tensor statistics
                arrays[f"{name}_sub4"] = np.ascontiguousarray(out[::4, ::4])
                arrays[f"{name}_rowmean"] = out.astype(np.float64).mean(axis=1).astype(np.float32)     # [h, 5]
                arrays[f"{name}_colmax"] = out.max(axis=0)                                              # [w, 5]

        def factory(ds, mp):
            e = tp.TorchParseNet(os.path.join(td, "parsenet.pt"), torch.device("cpu"), downsample=ds, max_mp=mp)
            return e
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            meta["adaptive"] = adaptive_cases(factory)
    out_dir = os.path.join(REPO, "tests", "golden")
    with open(os.path.join(out_dir, "parsenet.json"), "w", encoding="utf8") as f:
        json.dump(meta, f, indent=0)
    np.savez_compressed(os.path.join(out_dir, "parsenet.npz"), **arrays)
    print("adaptive cases:", [(c["calls"], round(c["net_downsample"], 3)) for c in meta["adaptive"]])


if __name__ == "__main__":
    main()
