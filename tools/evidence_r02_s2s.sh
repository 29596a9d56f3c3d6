cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_s2s; mkdir -p $O
timeout 600 python tools/s2s_bench.py 2048 512 4 3 20 > $O/s2s_bench.json 2> $O/s2s_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_s2s -o s2s -- python $R/tools/s2s_bench.py 2048 512 4 3 20 > $O/s2s_under_rocprof.json 2> $O/prof_s2s.err
cd $R
f=$(find $O/prof_s2s -name "*.db" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/s2s_kernel_stats.txt 2>&1
find $O -name "*.db" -size +20M -delete
cat $O/s2s_bench.json; head -16 $O/s2s_kernel_stats.txt
