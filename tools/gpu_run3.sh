cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
POCR_LSTM_STEP=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_c2_stepkernel.json 2> $O/bench_c2_stepkernel.err
python tools/stage_times.py 256 512 > $O/stage_alone_persist.txt 2>&1
POCR_LSTM_STEP=1 python tools/stage_times.py 256 512 > $O/stage_alone_step.txt 2>&1
timeout 600 python bench.py --workload c3 > $O/bench_c3.json 2> $O/bench_c3.err
POCR_FORCE_DIST=1 timeout 600 python bench.py --workload c3 > $O/bench_c3_rccl1.json 2> $O/bench_c3_rccl1.err
tail -5 $O/pytest.log; cut -c1-300 $O/bench_c2.json; echo; cut -c1-300 $O/bench_c2_stepkernel.json; echo; cat $O/stage_alone_persist.txt $O/stage_alone_step.txt; cut -c1-400 $O/bench_c3.json
