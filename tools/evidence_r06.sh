# Round-6 evidence set (one gpurun call): GPU test suite, the default bench line (with its extras), the bf16x3 / fp32-MFMA builds of the same
# bench, c3 (also over the RCCL transport), c4, c5, per-stage times alone, the store-hazard / masked-store probes, kernel traces of c2 / c4 / the seq2seq
# engine, PMC passes (HBM bytes, MFMA busy, LDS conflicts, instruction mix), cropper and layout-network benches.
cd $GRAFT_REPO_ROOT
export POCR_SOURCE_HEAD=$(cat gpurun_out/.source_head 2>/dev/null || cat .source_head 2>/dev/null)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.txt 2>&1; echo "rc $?" >> $O/smoke.txt
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
POCR_CONV_SPLIT=3 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_c2_bf16x3.json 2> $O/bench_c2_bf16x3.err
POCR_CONV_FP32=1 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_c2_fp32mfma.json 2> $O/bench_c2_fp32mfma.err
timeout 600 python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
POCR_FORCE_DIST=1 timeout 600 python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3_rccl_world1.json 2> $O/bench_c3_rccl_world1.err
timeout 600 python bench.py --workload c4 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 600 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
POCR_BENCH_HOST_CROPS=1 timeout 600 python bench.py --workload c5 > $O/bench_c5_host_crops.json 2> $O/bench_c5_host_crops.err
python tools/stage_times.py 256 512 > $O/stage_ms_c2_alone.txt 2>&1
python tools/stage_times.py 256 768 vgg_sa_ctc > $O/stage_ms_c4_alone.txt 2>&1
timeout 300 tools/bin/store_hazard_probe 4 > $O/store_hazard_probe.txt 2>&1
timeout 300 tools/bin/masked_store_probe 6 > $O/masked_store_probe.txt 2>&1
timeout 300 python tools/prof_stream_calls.py 6 3 > $O/stream_calls.txt 2>&1
timeout 300 python tools/crop_bench.py 80 20 > $O/crop_bench.txt 2>&1
timeout 300 python tools/parsenet_bench.py > $O/parsenet_bench.json 2>&1
timeout 600 python tools/s2s_bench.py 2048 512 4 3 20 > $O/s2s_bench.json 2> $O/s2s_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o c2 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c2_under_rocprof.json 2> $O/prof_c2.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o c4 -- python $R/bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c4_under_rocprof.json 2> $O/prof_c4.err
# (kernel trace with ONE decoding loop at a time: side by side the loops share the HBM and a kernel's duration is no longer its own)
POCR_S2S_DEPTH=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_s2s -o s2s -- python $R/tools/s2s_bench.py 2048 512 4 3 20 > $O/s2s_under_rocprof.json 2> $O/prof_s2s.err
for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc -o pmc_$name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$name.out 2> $O/pmc_$name.err
done
cd $R
for w in c2 c4 s2s; do f=$(find $O/prof_$w -name "*.db" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/${w}_kernel_stats.txt 2>&1; done
python tools/pmc_summary.py $(find $O/pmc -name "*.db") > $O/pmc_summary.json 2> $O/pmc_summary.err
python tools/s2s_roofline.py $O/s2s_kernel_stats.txt $O/s2s_bench.json 512 > $O/s2s_roofline.json 2> $O/s2s_roofline.err
find $O -name "*.db" -delete
tail -2 $O/pytest_gpu.txt; tail -2 $O/smoke.txt
for f in bench_c2 bench_c2_bf16x3 bench_c2_fp32mfma bench_c3 bench_c3_rccl_world1 bench_c4 bench_c5 bench_c5_host_crops; do echo $f; cut -c1-200 $O/$f.json; echo; done
head -16 $O/c2_kernel_stats.txt | cut -c1-200; cat $O/stage_ms_c2_alone.txt $O/stage_ms_c4_alone.txt; cat $O/s2s_roofline.json | cut -c1-600; cat $O/crop_bench.txt | tail -3
