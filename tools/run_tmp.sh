mkdir -p gpurun_out/r3k
timeout 1700 python -m pytest tests -x -q -m gpu -s > gpurun_out/r3k/pytest_gpu.txt 2>&1; echo "rc $?" >> gpurun_out/r3k/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r3k/bench_default.json 2> gpurun_out/r3k/bench_default.err
grep -E "^\[|\[rescale|\[c[0-9]|passed|failed|rc " gpurun_out/r3k/pytest_gpu.txt | tail -14; cut -c1-700 gpurun_out/r3k/bench_default.json
