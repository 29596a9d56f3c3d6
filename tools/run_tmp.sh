mkdir -p gpurun_out/r3q
timeout 1200 python -m pytest tests/test_parsenet.py tests/test_crop.py tests/test_pipeline.py -x -q -m gpu > gpurun_out/r3q/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/r3q/pytest.txt
tail -8 gpurun_out/r3q/pytest.txt
