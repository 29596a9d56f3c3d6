#!/bin/bash
# PMC passes over one run of tools/bin/conv_wino_bench (direct kernel + Winograd kernel of one layer); prints per kernel and counter the mean per dispatch
# usage (on the GPU box): tools/wino_pmc.sh <out dir> <bench binary> <layer> [more bench args]
mkdir -p $1; O=$(realpath $1); B=$(realpath $2); shift 2
R=$(pwd); mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc$i -o p$i -- $B "$@" > $O/pmc$i.out 2> $O/pmc$i.err || echo "pass $i failed: $(tail -2 $O/pmc$i.err)"
done
cd $R
python3 - $O <<'PY'
import sqlite3, sys, glob, collections
out = collections.defaultdict(dict)
for db in sorted(glob.glob(sys.argv[1] + "/pmc*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    try:
        rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print("no counters in", db, e); continue
    for kern, ctr, total, nd in rows:
        short = "wino" if "wino" in kern else "direct" if "conv3x3_bf16x3" in kern else kern[:20]
        out[short].setdefault(ctr, total / max(1, nd))
for k, c in out.items():
    print("==", k)
    for name in sorted(c): print(f"  {name:40s} {c[name]:.4g}")
PY
find $O -name "*.db" -delete
