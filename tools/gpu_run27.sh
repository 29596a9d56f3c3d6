cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2zz; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in conv_bench_bf16 conv_bench_bf16_n; do
  timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_$b -o p -- $R/tools/bin/$b 9 > /dev/null 2> $O/$b.err
  cd $R; python tools/pmc_summary.py $(find $O/pmc_$b -name "*.db") > $O/$b.json; cd /tmp
done
cd $R
python - <<'PY'
import json
for b in ('conv_bench_bf16','conv_bench_bf16_n'):
    d=json.load(open(f'gpurun_out/r2zz/{b}.json'))
    for k,v in d.items():
        if 'bf16x3' in k and '5, 1, 2, 1' in k: print(b, k[28:100], round(v.get('lds_bank_conflict_share',0),3), round(v.get('lds_util',0),3))
PY
