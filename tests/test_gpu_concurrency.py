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
