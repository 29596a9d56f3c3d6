#!/bin/bash
mkdir -p gpurun_out/r2o
for t in 147456 98304 73728 49152 32768; do
  echo "== target $t" >> gpurun_out/r2o/c5_targets.txt
  POCR_LAUNCH_TARGET=$t timeout 600 python bench.py --workload c5 --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_page'])" >> gpurun_out/r2o/c5_targets.txt
done
cat gpurun_out/r2o/c5_targets.txt
