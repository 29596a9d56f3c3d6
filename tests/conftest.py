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
        return self.meta[k]

    @property
    def n(self):
        return len(self.meta["widths"])

    def spec(self):
        from pero_ocr_amd import netspec
        return netspec.NetSpec.from_json(self.meta["spec"])

    def weights(self):
        from pero_ocr_amd import netspec
        extra = {"boundary_bias": self.meta["boundary_bias"]} if "boundary_bias" in self.meta else {}
        return netspec.generate_weights(self.spec(), self.meta["weight_seed"], **extra)

    def crops(self):
        from pero_ocr_amd import synth
        return synth.make_crops(self.meta["crop_seed"], self.meta["widths"], self.meta["height"])

    def argmax(self, i):
        return self.arrays[f"argmax_{i}"].astype(np.int64)

    def margin(self, i):
        """Reference top-2 logit margin per frame (how robust 'argmax identical' is on that frame)."""
        return self.arrays[f"margin_{i}"]

    def write_engine_json(self, tmpdir):
        """Engine JSON in the reference's schema + the build-specific "net" key."""
        path = os.path.join(str(tmpdir), "ocr.json")
        with open(path, "w", encoding="utf8") as f:
            json.dump({"line_px_height": self.meta["height"], "line_vertical_scale": 1.0,
                       "checkpoint": "absent.pocrw", "characters": self.meta["characters"][:-1],
                       "net_name": "VGG_BLSTM_CTC",
                       "net": {"arch": self.meta["spec"].get("arch", "vgg_blstm_ctc"),
                               "weight_seed": self.meta["weight_seed"]}}, f)
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
