cd $GRAFT_REPO_ROOT
timeout 900 python tools/train_step_bench.py h 7b 2 bf16 2>&1 < /dev/null | tail -1
timeout 900 python tools/train_step_bench.py b 2b 2 bf16 2>&1 < /dev/null | tail -1
