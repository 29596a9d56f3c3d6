"""The Winograd F(2,3) experiment (tools/experiments/conv_wino.hpp; NOT part of the library - DESIGN.md section 9): its bench tool
checks the kernel against a float64 CPU reference on sampled outputs and against the shipped direct kernel on every output.  The
test runs the tool on a small case per layer shape and holds the numbers it prints to the layer-level bars the direct kernel meets
(profiles/r05_winograd.txt).  Skipped when the tool was not built (tools/build_wino_variants.sh base:)."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(REPO, "tools", "bin", "conv_wino_bench")


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [9, 8, 6, 4, 7, 3])
def test_winograd_kernel_against_float64_and_direct(layer):
    if not os.path.exists(TOOL):
        pytest.skip("tools/bin/conv_wino_bench not built")
    out = subprocess.run([TOOL, str(layer), "6", "576"], capture_output=True, text=True, timeout=300).stdout
    line = next(l for l in out.splitlines() if "Winograd F(2,3)" in l)
    ref_max = float(re.search(r"vs float64: max ([0-9.e+-]+)", line).group(1))
    diff = float(re.search(r"max\|wino - direct\| ([0-9.e+-]+) \(NaN (\d+)\)", line).group(1))
    nan = int(re.search(r"\(NaN (\d+)\)", line).group(1))
    direct = next(l for l in out.splitlines() if "direct f16x2 P2 (shipped)" in l)
    direct_max = float(re.search(r"vs float64: max ([0-9.e+-]+)", direct).group(1))
    assert nan == 0
    assert ref_max < 1e-6 and ref_max < 2.0 * direct_max + 1e-7, (ref_max, direct_max)      # outputs are O(1): fp32-level agreement with float64
    assert diff < 5e-6, diff
