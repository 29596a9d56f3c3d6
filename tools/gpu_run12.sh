cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k; mkdir -p $O
for l in 2 3 4 5 6 7 8 9; do timeout 120 tools/bin/conv_bench_bf16 $l 256 576 2>&1; done > $O/conv_bf16x3_bench.txt; cut -c1-175 $O/conv_bf16x3_bench.txt | grep -v "2x2\|M-split   \|5x48\|NT256 N" 
python tools/stage_times.py 256 512 2>&1 | tail -1
timeout 600 python tools/c3_worst.py c3 > $O/c3_err_bf16x3_v3.txt 2>&1; tail -5 $O/c3_err_bf16x3_v3.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 500 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-220 $O/bench_c2.json
