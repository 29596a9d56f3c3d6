#!/bin/bash
mkdir -p gpurun_out/r2u
for m in 1 2 4; do
  export POCR_LSTM_MULTI=$m
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "engine_matches or c2_full" 2>&1 | tail -1
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2u/bench_c2_multi$m.json 2> gpurun_out/r2u/bench_c2_multi$m.err
  python -c "import json; d=json.load(open('gpurun_out/r2u/bench_c2_multi$m.json')); print('multi', $m, d['value'], d['ms_per_step'], d['stage_ms'])"
  python tools/stage_times.py 256 512 2>&1 | tail -1 | cut -c150-260
done
