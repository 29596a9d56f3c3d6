cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
python tools/stage_times.py 256 512 > $O/stage_alone.txt 2>&1; tail -1 $O/stage_alone.txt
timeout 500 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-220 $O/bench_c2.json
POCR_CONV_FP32=1 timeout 500 python bench.py --no-cpu-baseline > $O/bench_c2_fp32.json 2> $O/bench_c2_fp32.err; cut -c1-220 $O/bench_c2_fp32.json
