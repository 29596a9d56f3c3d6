#!/bin/bash
mkdir -p gpurun_out/r2q
for l in 9 8 6 7 5 4 3 2; do timeout 300 tools/bin/conv_bench_bf16 $l >> gpurun_out/r2q/conv_bf16_direct.txt 2>&1; done
grep -v "^$" gpurun_out/r2q/conv_bf16_direct.txt | cut -c1-150
