cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --workload c3 > $O/bench_c3.json 2> $O/bench_c3.err
POCR_FORCE_DIST=1 timeout 600 python bench.py --workload c3 > $O/bench_c3_rccl1.json 2> $O/bench_c3_rccl1.err
timeout 500 python bench.py --workload c4 > $O/bench_c4.json 2> $O/bench_c4.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o r2 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2> $O/prof_c2.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o r2 -- python $R/bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c4_under_rocprof.json 2> $O/prof_c4.err
cd $R
for w in c2 c4; do f=$(find $O/prof_$w -name "*.db" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/${w}_kernel_stats.txt 2>&1; done
find $O -name "*.db" -size +30M -delete
tail -4 $O/pytest.log; cut -c1-250 $O/bench_c2.json; echo; cut -c1-250 $O/bench_c3.json; echo; head -12 $O/c2_kernel_stats.txt
