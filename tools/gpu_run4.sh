cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 600 python tools/c3_worst.py c3 > $O/c3_worst_persist.txt 2>&1
POCR_LSTM_STEP=1 timeout 600 python tools/c3_worst.py c3 > $O/c3_worst_step.txt 2>&1
cat $O/c3_worst_persist.txt $O/c3_worst_step.txt
