cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
timeout 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --workload c3 > $O/bench_c3.json 2> $O/bench_c3.err
POCR_FORCE_DIST=1 timeout 600 python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3_rccl_world1.json 2> $O/bench_c3_rccl_world1.err
timeout 500 python bench.py --workload c4 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 500 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
python tools/stage_times.py 256 512 > $O/stage_alone_c2.txt 2>&1
python tools/stage_times.py 256 768 vgg_sa_ctc > $O/stage_alone_c4.txt 2>&1
timeout 300 python tools/crop_bench.py 80 20 > $O/crop_bench.txt 2>&1
timeout 300 python tools/parsenet_bench.py > $O/parsenet_bench.json 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o r2 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2> $O/prof_c2.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o r2 -- python $R/bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c4_under_rocprof.json 2> $O/prof_c4.err
for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc -o pmc_$name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$name.out 2> $O/pmc_$name.err
done
cd $R
f=$(find $O/prof_c2 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/c2_kernel_stats.txt 2>&1
f=$(find $O/prof_c4 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/c4_kernel_stats.txt 2>&1
python tools/pmc_summary.py $(find $O/pmc -name "*.db") > $O/pmc_summary.json 2> $O/pmc_summary.err
find $O -name "*.db" -size +20M -delete
tail -2 $O/pytest_gpu.txt
for f in bench_c2 bench_c3 bench_c3_rccl_world1 bench_c4 bench_c5; do cut -c1-180 $O/$f.json; echo; done; head -8 $O/c2_kernel_stats.txt; tail -1 $O/stage_alone_c2.txt
