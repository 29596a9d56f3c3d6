mkdir -p gpurun_out/r3w
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r3w/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r3w/pytest.txt
tail -4 gpurun_out/r3w/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 auto', r['value'], r['ms_per_step'], r['stage_ms'])"
for sl in 0 2 4; do echo "c3 SL=$sl"; POCR_LSTM_SL=$sl timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])"; done
echo "c5 auto"; timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r.get('page_at_a_time'))"
