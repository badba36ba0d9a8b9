cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 < /dev/null | grep -E "passed|failed|error" | tail -3
