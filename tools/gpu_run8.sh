cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_parsenet.py -x -q -m gpu > $O/pytest_parsenet.log 2>&1; tail -15 $O/pytest_parsenet.log
timeout 300 python tools/parsenet_bench.py > $O/parsenet_bench.json 2>&1; cat $O/parsenet_bench.json
python tools/stage_times.py 256 768 vgg_sa_ctc > $O/stage_alone_c4.txt 2>&1; tail -1 $O/stage_alone_c4.txt
