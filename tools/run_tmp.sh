mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_crop.py tests/test_pipeline.py -x -q -m gpu > gpurun_out/r3j/pytest_crop.txt 2>&1; echo "rc $?" >> gpurun_out/r3j/pytest_crop.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c5 > gpurun_out/r3j/bench_c5_resident.json 2> gpurun_out/r3j/bench_c5_resident.err
POCR_BENCH_HOST_CROPS=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --workload c5 > gpurun_out/r3j/bench_c5_hostcrops.json 2> gpurun_out/r3j/bench_c5_hostcrops.err
tail -6 gpurun_out/r3j/pytest_crop.txt
for f in resident hostcrops; do python - <<PY
import json
r=json.loads(open('gpurun_out/r3j/bench_c5_$f.json').read().strip().splitlines()[-1])
print('$f', r['value'], r['ms_per_step'], r.get('page_at_a_time'))
PY
done
tail -3 gpurun_out/r3j/bench_c5_resident.err
