mkdir -p gpurun_out/r3d
timeout 1700 python -m pytest tests -x -q -m gpu -s > gpurun_out/r3d/pytest_gpu.txt 2>&1; echo "rc $?" >> gpurun_out/r3d/pytest_gpu.txt
( time timeout 900 python bench.py > gpurun_out/r3d/bench_default.json 2> gpurun_out/r3d/bench_default.err ) 2>> gpurun_out/r3d/bench_default.err
grep -E "^\[c|passed|failed|rc " gpurun_out/r3d/pytest_gpu.txt | tail -12; cat gpurun_out/r3d/bench_default.json | cut -c1-3000; tail -5 gpurun_out/r3d/bench_default.err
