#!/bin/bash
mkdir -p gpurun_out/r2t
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2t/pytest_gpu.txt 2>&1; echo "rc $?" >> gpurun_out/r2t/pytest_gpu.txt
tail -3 gpurun_out/r2t/pytest_gpu.txt
for narrow in 0 1; do
  POCR_LSTM_NARROW=$narrow timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2t/bench_c2_narrow$narrow.json 2> gpurun_out/r2t/bench_c2_narrow$narrow.err
  python -c "import json; d=json.load(open('gpurun_out/r2t/bench_c2_narrow$narrow.json')); print('narrow', $narrow, d['value'], d['ms_per_step'], d['stage_ms'])"
  POCR_LSTM_NARROW=$narrow python tools/stage_times.py 256 512 2>&1 | tail -2
done
POCR_LSTM_NARROW=0 timeout 600 python bench.py --workload c3 --no-cpu-baseline 2>/dev/null | cut -c1-200
POCR_LSTM_NARROW=1 timeout 600 python bench.py --workload c3 --no-cpu-baseline 2>/dev/null | cut -c1-200
