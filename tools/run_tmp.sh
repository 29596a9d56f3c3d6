mkdir -p gpurun_out/r3y
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resident_recurrence or c2_full or ragged or c3" > gpurun_out/r3y/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r3y/pytest.txt
tail -3 gpurun_out/r3y/pytest.txt
timeout 300 python tools/stage_times.py 256 512 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', r['value'], r['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', r['value'], r['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', r['value'], r['ms_per_step'], r.get('page_at_a_time'))"
timeout 600 python tools/c5_ocr_stages.py 2>&1 | grep -v WARNING | grep "slot [01]"
