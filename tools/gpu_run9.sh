cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout 600 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err; cat $O/bench_c5.json; tail -3 $O/bench_c5.err
python tools/crop_bench.py 80 > $O/crop_bench.json 2>&1; cat $O/crop_bench.json
