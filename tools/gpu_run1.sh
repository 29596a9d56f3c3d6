cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q -k "not c3" > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2a/pytest.log
timeout 300 python bench.py > gpurun_out/r2a/bench_c2.json 2> gpurun_out/r2a/bench_c2.err
POCR_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2a/bench_c2_rccl1.json 2> gpurun_out/r2a/bench_c2_rccl1.err
timeout 300 python bench.py --workload c4 > gpurun_out/r2a/bench_c4.json 2> gpurun_out/r2a/bench_c4.err
timeout 120 python bench.py --gpus 2 --no-cpu-baseline > gpurun_out/r2a/bench_gpus2.out 2>&1; echo "rc $?" >> gpurun_out/r2a/bench_gpus2.out
tail -5 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/bench_c2.json | cut -c1-600
