#!/bin/bash
mkdir -p gpurun_out/r2x
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2x/pytest_gpu.txt 2>&1; echo "rc $?" >> gpurun_out/r2x/pytest_gpu.txt
tail -4 gpurun_out/r2x/pytest_gpu.txt | cut -c1-300
for w in c2 c4; do timeout 600 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['stage_ms'], d.get('encoder'))"; done
python tools/stage_times.py 256 512 2>&1 | tail -1
python tools/stage_times.py 256 768 vgg_sa_ctc 2>&1 | tail -1
