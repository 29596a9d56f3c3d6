#!/bin/bash
mkdir -p gpurun_out/r2p
for k in 1 2 4 8; do
  timeout 600 python bench.py --workload c5 --steps 16 --warmup 2 --pages-per-batch $k > gpurun_out/r2p/c5_ppb$k.json 2> gpurun_out/r2p/c5_ppb$k.err
  python -c "import sys,json; d=json.loads(open('gpurun_out/r2p/c5_ppb$k.json').read()); print($k, d['value'], d['ms_per_step'], d['page_at_a_time'])" || tail -5 gpurun_out/r2p/c5_ppb$k.err
done
