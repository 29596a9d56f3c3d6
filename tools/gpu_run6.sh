cd $GRAFT_REPO_ROOT
POCR_LSTM_DBG=1 POCR_LSTM_Z=1 python tools/stage_times.py 256 512 2>&1 | grep -B34 "lstm dbg" | tail -36
