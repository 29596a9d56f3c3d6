# Round-3 evidence set (one gpurun call): kernel trace + PMC passes of the default bench, per-stage times alone.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o c2 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c2_under_rocprof.json 2> $O/prof_c2.err
for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc -o pmc_$name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$name.out 2> $O/pmc_$name.err
done
cd $R
python tools/rocprof_summary.py $(find $O/prof_c2 -name "*.db" | head -1) > $O/bench_c2_kernel_stats.txt 2> $O/kernel_stats.err
python tools/pmc_summary.py $(find $O/pmc -name "*.db") > $O/pmc_summary.json 2> $O/pmc_summary.err
find $O -name "*.db" -size +20M -delete
python tools/stage_times.py 256 512 > $O/stage_ms_c2_alone.txt 2>&1
python tools/stage_times.py 256 768 vgg_sa_ctc > $O/stage_ms_c4_alone.txt 2>&1
head -30 $O/bench_c2_kernel_stats.txt | cut -c1-230
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/pmc_summary.json'))
for k,v in d.items():
    if 'conv3x3' in k or 'lstm' in k or 'conv1' in k:
        print(k[28:120], {kk: round(vv,3) for kk,vv in v.items() if kk in ('mfma_util','lds_bank_conflict_share','lds_util')}, 'GB', round(v.get('hbm_bytes_per_launch',0)/1e9,2), 'rd', round(v.get('hbm_read_bytes_corrected',0)/1e9,2))
PY
cat $O/stage_ms_c2_alone.txt $O/stage_ms_c4_alone.txt
