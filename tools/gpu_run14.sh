#!/bin/bash
# resident cropper: parity tests + bench
mkdir -p gpurun_out/r2m
timeout 600 python -m pytest tests/test_crop.py -x -q -m gpu > gpurun_out/r2m/pytest_crop.txt 2>&1; echo "rc $?" >> gpurun_out/r2m/pytest_crop.txt
timeout 300 python tools/crop_bench.py 80 10 > gpurun_out/r2m/crop_bench.txt 2>&1
timeout 300 python tools/crop_bench.py 47 10 >> gpurun_out/r2m/crop_bench.txt 2>&1
tail -5 gpurun_out/r2m/pytest_crop.txt; cat gpurun_out/r2m/crop_bench.txt
