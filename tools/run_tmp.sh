mkdir -p gpurun_out/r3n
timeout 300 python tools/stage_times.py 256 512 > gpurun_out/r3n/stage_c2.txt 2>&1
timeout 300 python tools/stage_times.py 256 768 vgg_sa_ctc > gpurun_out/r3n/stage_c4.txt 2>&1
POCR_CONV_SPLIT=3 timeout 300 python tools/stage_times.py 256 512 > gpurun_out/r3n/stage_c2_bf16x3.txt 2>&1
timeout 300 python tools/parsenet_bench.py > gpurun_out/r3n/parsenet.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_parsenet.py -x -q -m gpu > gpurun_out/r3n/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r3n/pytest.txt
cat gpurun_out/r3n/stage_c2.txt gpurun_out/r3n/stage_c4.txt gpurun_out/r3n/stage_c2_bf16x3.txt; tail -3 gpurun_out/r3n/parsenet.txt; tail -3 gpurun_out/r3n/pytest.txt
