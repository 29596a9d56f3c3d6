cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lstm_persistent or layerwise_parity or c2_full" > $O/pytest_lstm.log 2>&1; echo "rc $?" >> $O/pytest_lstm.log
tail -4 $O/pytest_lstm.log
for z in 1 2 4; do POCR_LSTM_Z=$z python tools/stage_times.py 256 512 > $O/stage_alone_z$z.txt 2>&1; cat $O/stage_alone_z$z.txt; done
for z in 1 2 4; do POCR_LSTM_Z=$z timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2_z$z.json 2> $O/bench_c2_z$z.err; cut -c1-260 $O/bench_c2_z$z.json; echo; done
