"""Fresh engines under load (tools/stress_first_job.py): the first call of a new engine must equal the single-threaded result while
other engines of the process and a layout network keep the GPU busy.  Regression test for the race between the padding-column
constants' device-to-device copy and the first launch (round 4); it is a timing test - it caught that race in ~1 of 100 engines."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_first_call_of_fresh_engines_under_load():
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "stress_first_job.py"), "12", "3"], capture_output=True, text=True, timeout=300)
    last = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and last, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.gpu
def test_three_launches_in_flight_on_fresh_engines_are_bit_identical():
    """The activations of a launch do not depend on what the other two slots run at the same time - checked on FRESH engines, where a
    store of a convolution that was masked by an out-of-range buffer offset (not by a branch) once left wrong low planes in another
    launch's activations: only with three launches in flight, only in the first round after pocr_create (tools/three_in_flight.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("three_in_flight", os.path.join(REPO, "tools", "three_in_flight.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(4) == 0
