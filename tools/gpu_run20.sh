#!/bin/bash
# N-rank control flow on one GPU (test hook): RCCL refuses duplicate devices -> gloo fallback
mkdir -p gpurun_out/r2s
export POCR_BENCH_SHARE_GPU=1 POCR_RCCL_INIT_TIMEOUT=60
timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2s/c2_2ranks.json 2> gpurun_out/r2s/c2_2ranks.err; echo "rc $?"
tail -3 gpurun_out/r2s/c2_2ranks.err; cut -c1-300 gpurun_out/r2s/c2_2ranks.json
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --workload c3 > gpurun_out/r2s/c3_2ranks.json 2> gpurun_out/r2s/c3_2ranks.err; echo "rc $?"
tail -3 gpurun_out/r2s/c3_2ranks.err; cut -c1-300 gpurun_out/r2s/c3_2ranks.json
