cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 < /dev/null | grep -E "passed|failed|Error" | tail -3
