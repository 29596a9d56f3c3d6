cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lstm_persistent or layerwise_parity" > gpurun_out/r2b/pytest_lstm.log 2>&1; echo "rc $?" >> gpurun_out/r2b/pytest_lstm.log
tail -30 gpurun_out/r2b/pytest_lstm.log
