#!/bin/bash
mkdir -p gpurun_out/check
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/check/pytest_gpu.txt 2>&1; echo "rc $?" >> gpurun_out/check/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/check/bench_c2.json 2> gpurun_out/check/bench_c2.err
tail -4 gpurun_out/check/pytest_gpu.txt; cat gpurun_out/check/bench_c2.json | cut -c1-600
