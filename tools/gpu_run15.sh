#!/bin/bash
mkdir -p gpurun_out/r2n
timeout 600 python -m pytest tests/test_crop.py tests/test_pipeline.py -x -q -m gpu > gpurun_out/r2n/pytest_crop.txt 2>&1; echo "rc $?" >> gpurun_out/r2n/pytest_crop.txt
timeout 300 python tools/crop_bench.py 80 20 > gpurun_out/r2n/crop_bench.txt 2>&1
timeout 300 python tools/crop_bench.py 47 20 >> gpurun_out/r2n/crop_bench.txt 2>&1
timeout 600 python bench.py --workload c5 --steps 10 --warmup 2 > gpurun_out/r2n/bench_c5.json 2> gpurun_out/r2n/bench_c5.err
tail -3 gpurun_out/r2n/pytest_crop.txt; cat gpurun_out/r2n/crop_bench.txt gpurun_out/r2n/bench_c5.json; tail -3 gpurun_out/r2n/bench_c5.err
