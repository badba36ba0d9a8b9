cd $GRAFT_REPO_ROOT
timeout 1200 python tools/train_step_bench.py h 7b 2 2>&1 < /dev/null | tail -2
