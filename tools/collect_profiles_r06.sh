# copies the summaries of tools/evidence_r06.sh (gpurun_out/r06, scratch) into profiles/ (tracked) under their round-4 names
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r06; P=profiles
for f in bench_c2 bench_c2_bf16x3 bench_c2_fp32mfma bench_c2_under_rocprof bench_c3 bench_c3_rccl_world1 bench_c4 bench_c4_under_rocprof bench_c5 bench_c5_host_crops s2s_bench parsenet_bench pmc_summary s2s_roofline; do
  [ -s $O/$f.json ] && cp $O/$f.json $P/r06_$f.json
done
cp $O/c2_kernel_stats.txt $P/r06_bench_c2_kernel_stats.txt
cp $O/c4_kernel_stats.txt $P/r06_bench_c4_kernel_stats.txt
cp $O/s2s_kernel_stats.txt $P/r06_s2s_kernel_stats.txt
cp $O/stage_ms_c2_alone.txt $P/r06_stage_ms_c2_single_chunk_alone.txt
cp $O/stage_ms_c4_alone.txt $P/r06_stage_ms_c4_single_chunk_alone.txt
cp $O/crop_bench.txt $P/r06_crop_bench_resident.txt
# (r06_default_call_stream.txt carries a hand-written appendix: refreshed by hand from $O/stream_calls.txt, not overwritten here)
grep -oE "\[c[0-9a-z ]*\].*|\[rescale.*|\[range.*|\[smoke.*|\[full.*|\[host.*|\[s2s range.*|^[0-9]+ passed.*|^[0-9]+ failed.*" $O/pytest_gpu.txt > $P/r06_parity_prints.txt || true
ls $P | grep r06_ | wc -l
