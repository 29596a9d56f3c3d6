"""The N-rank path of bench.py executed by the suite: `bench.py --gpus 2` starts its two ranks itself
(rank_launch_plan -> torch.distributed.run -> unique-id rendezvous -> pocr_comm_init).  On the single-GPU test box both ranks
share GPU 0 (test hook POCR_BENCH_SHARE_GPU=1): RCCL refuses a communicator with a duplicate device, so the ranks must agree
on the gloo carrier, run the sharded workload to the end and print ONE line that cannot be read as an RCCL result -
`value` null, the measured figure under `value_gloo_fallback`, `rccl_ranks` 0.  On an 8-GPU node the same control flow gets
its communicator; what this test pins is everything around it."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _run(argv, timeout=900):
    env = dict(os.environ, POCR_BENCH_SHARE_GPU="1", POCR_RCCL_INIT_TIMEOUT="60", OMP_NUM_THREADS="4")
    env.pop("POCR_BENCH_REQUIRE_RCCL", None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"exactly one JSON line on stdout, got {len(lines)}: {p.stdout[:2000]}"
    return json.loads(lines[0]), p.stderr


@pytest.mark.gpu
@pytest.mark.timeout(1000)
def test_two_ranks_c3_stream_end_to_end_on_the_gloo_fallback():
    r, err = _run(["--gpus", "2", "--workload", "c3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["steps"] == 1
    assert r["rccl_ranks"] == 0 and r["value"] is None and r["value_gloo_fallback"] > 0
    assert r["config"]["collective"].startswith("gloo FALLBACK")
    # (the c3 branch asserts texts == the reference fixture's transcriptions before it prints: rc 0 is that check)
    assert "transcriptions checked against the reference fixture" in r["config"]["workload"]
    assert "2 rank(s)" in r["config"]["workload"] and "gloo carrier (NOT RCCL)" in r["config"]["workload"]
    assert "gloo FALLBACK" in err


@pytest.mark.gpu
@pytest.mark.timeout(1000)
def test_eight_ranks_deal_the_c3_stream_and_reproduce_the_fixture():
    """The size the driver's scaling run ends with: eight ranks (all on GPU 0 here), the 343 reference chunks of the c3 stream dealt to them,
    one all-gather per pass, every transcription equal to the reference fixture (asserted by the bench on every rank before it prints)."""
    r, _err = _run(["--gpus", "8", "--workload", "c3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 0 and r["value"] is None and r["value_gloo_fallback"] > 0
    assert "8 rank(s)" in r["config"]["workload"] and "transcriptions checked against the reference fixture" in r["config"]["workload"]


@pytest.mark.gpu
@pytest.mark.timeout(1000)
def test_two_ranks_c2_weak_scaling_step_loop_on_the_gloo_fallback():
    r, _err = _run(["--gpus", "2", "--workload", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["lines_per_step"] == 512
    assert r["rccl_ranks"] == 0 and r["value"] is None and r["value_gloo_fallback"] > 0
    assert "cpu_baseline" not in r and "extra" not in r


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_refuse_the_fallback_when_rccl_is_required():
    env = dict(os.environ, POCR_BENCH_SHARE_GPU="1", POCR_RCCL_INIT_TIMEOUT="60", POCR_BENCH_REQUIRE_RCCL="1", OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "c2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "POCR_BENCH_REQUIRE_RCCL=1" in p.stderr
