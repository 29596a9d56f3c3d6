import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """A fixture written by oracle/gen_golden.py: expected outputs of the imported reference
    engine; the inputs (crops, weights) are regenerated from the stored seeds."""

    def __init__(self, name):
        with open(os.path.join(GOLDEN_DIR, f"{name}.json"), encoding="utf8") as f:
            self.meta = json.load(f)
        self.arrays = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
        self.name = name

    def __getattr__(self, k):
        try:
            return self.__dict__["meta"][k]
        except KeyError:
            raise AttributeError(k) from None

    @property
    def n(self):
        return len(self.meta["widths"])

    def spec(self):
        from pero_ocr_amd import netspec
        return netspec.NetSpec.from_json(self.meta["spec"])

    def weights(self):
        """Seeded weights + the data-calibrated tensors the fixture stores ("override_<name>", see
        oracle/gen_golden.py calibrated_weights)."""
        from pero_ocr_amd import netspec
        extra = dict(self.meta.get("weight_kwargs", {}))
        if "boundary_bias" in self.meta:
            extra["boundary_bias"] = self.meta["boundary_bias"]
        w = netspec.generate_weights(self.spec(), self.meta["weight_seed"], **extra)
        for k in self.arrays.files:
            if k.startswith("override_"):
                w[k[len("override_"):]] = self.arrays[k]
        return w

    def crops(self):
        from pero_ocr_amd import synth
        return synth.make_crops(self.meta["crop_seed"], self.meta["widths"], self.meta["height"],
                                self.meta.get("crop_indices"))

    def _frame_slice(self, i):
        if not hasattr(self, "_row_off"):
            self._row_off = np.concatenate([[0], np.cumsum(self.arrays["shapes"][:, 0])])
        return slice(int(self._row_off[i]), int(self._row_off[i + 1]))

    def _whole(self, key):
        """npz members are decompressed on every access: keep the concatenated ones."""
        if not hasattr(self, "_memo"):
            self._memo = {}
        if key not in self._memo:
            self._memo[key] = self.arrays[key]
        return self._memo[key]

    def argmax(self, i):
        if "argmax_all" in self.arrays.files:
            return self._whole("argmax_all")[self._frame_slice(i)].astype(np.int64)
        return self.arrays[f"argmax_{i}"].astype(np.int64)

    def margin(self, i):
        """Reference top-2 logit margin per frame (how robust 'argmax identical' is on that frame)."""
        if "margin_all" in self.arrays.files:
            return self._whole("margin_all")[self._frame_slice(i)]
        return self.arrays[f"margin_{i}"]

    def rows(self, i):
        """Reference logits of the sampled frames sample_rows[i] of line i: [k, C]."""
        if "rows_all" in self.arrays.files:
            if not hasattr(self, "_srow_off"):
                self._srow_off = np.concatenate([[0], np.cumsum([len(r) for r in self.meta["sample_rows"]])])
            return self._whole("rows_all")[int(self._srow_off[i]):int(self._srow_off[i + 1])]
        return self.arrays[f"rows_{i}"]

    def rows64(self, i):
        """The sampled rows computed in float64 by the restated network (oracle/gen_truth_rows.py), or None."""
        if "rows_all" not in self.arrays.files:
            return None
        r = self.rows(i)                 # builds _srow_off
        a, b = int(self._srow_off[i]), int(self._srow_off[i + 1])
        if "rows64_delta16" in self.arrays.files:      # truth = reference row - float16(reference - truth), exact to < 1e-6
            return (r.astype(np.float64) - self._whole("rows64_delta16")[a:b].astype(np.float64)).astype(np.float32)
        if "rows64_all" not in self.arrays.files:
            return None
        return self._whole("rows64_all")[a:b]

    def l2(self, i):
        if "l2_all" in self.arrays.files:
            return float(self._whole("l2_all")[i])
        return float(self.arrays[f"l2_{i}"][0])

    def rowlse(self, i):
        return self._whole("rowlse")[self._frame_slice(i)]

    def write_engine_json(self, tmpdir):
        """Engine JSON in the reference's schema + the build-specific "net" key."""
        path = os.path.join(str(tmpdir), "ocr.json")
        checkpoint = "absent.pocrw"
        if self.meta.get("weight_kwargs") or any(k.startswith("override_") for k in self.arrays.files):
            from pero_ocr_amd import netspec        # not reproducible from the seed alone: ship a weight blob
            checkpoint = "weights.pocrw"
            netspec.save_blob(os.path.join(str(tmpdir), checkpoint), self.spec(), self.weights())
        cfg = {"line_px_height": self.meta["height"], "line_vertical_scale": 1.0,
               "checkpoint": checkpoint, "characters": self.meta["characters"][:-1],
               "net_name": "VGG_BLSTM_CTC",
               "net": {"arch": self.meta["spec"].get("arch", "vgg_blstm_ctc"),
                       "weight_seed": self.meta["weight_seed"]}}
        if "embed_id" in self.meta:                       # the reference's own keys (line_ocr_engine.py:32-42)
            cfg.update(embed_num=self.meta["embed_num"], embed_id=self.meta["embed_id"])
        with open(path, "w", encoding="utf8") as f:
            json.dump(cfg, f)
        return path


_cache = {}


@pytest.fixture
def golden():
    def get(name):
        if name not in _cache:
            _cache[name] = Golden(name)
        return _cache[name]
    return get


def gpu_available():
    try:
        from pero_ocr_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False
