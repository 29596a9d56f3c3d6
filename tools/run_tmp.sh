mkdir -p gpurun_out/r3h
for c in 0 1 2 3 4; do echo "AGG_CFG=$c" >> gpurun_out/r3h/agg.txt; POCR_AGG_CFG=$c timeout 300 python tools/stage_times.py 256 512 >> gpurun_out/r3h/agg.txt 2>&1; done
cat gpurun_out/r3h/agg.txt
