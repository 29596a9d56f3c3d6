mkdir -p gpurun_out/r3x
timeout 1700 python -m pytest tests -x -q -m gpu -s > gpurun_out/r3x/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r3x/pytest.txt
grep -E "\[c[0-9u]+\]|\[rescale|passed|failed|rc " gpurun_out/r3x/pytest.txt | sed 's/^\.*//' | tail -14
timeout 300 python tools/stage_times.py 256 512 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', r['value'], r['ms_per_step'])"
POCR_LSTM_FP32=1 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 fp32 recurrence', r['value'], r['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', r['value'], r['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', r['value'], r['ms_per_step'], r.get('page_at_a_time'))"
POCR_LSTM_FP32=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 fp32 recurrence', r['value'], r['ms_per_step'], r.get('page_at_a_time'))"
