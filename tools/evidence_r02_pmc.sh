cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc -o pmc_$name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$name.out 2> $O/pmc_$name.err
done
cd $R
python tools/pmc_summary.py $(find $O/pmc -name "*.db") > $O/pmc_summary.json 2> $O/pmc_summary.err
find $O -name "*.db" -size +20M -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_pmc/pmc_summary.json'))
for k,v in d.items():
    if 'conv3x3' in k or 'lstm' in k or 'conv1' in k:
        print(k[28:95], {kk: round(vv,3) for kk,vv in v.items() if kk in ('mfma_util','lds_bank_conflict_share','lds_util')}, round(v.get('hbm_bytes_per_launch',0)/1e9,2))
PY
