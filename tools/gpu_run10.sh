cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
DUMP=1 timeout 120 tools/bin/conv_bench_bf16 6 16 576 2>&1 | grep "A-plane"
for l in 9 6 4 2; do timeout 120 tools/bin/conv_bench_bf16 $l 256 576 > $O/bf16_conv$l.txt 2>&1; cat $O/bf16_conv$l.txt; done
