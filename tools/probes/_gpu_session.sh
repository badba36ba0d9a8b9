cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rmsnorm_prologue" 2>&1 < /dev/null | grep -E "passed|failed|Error|assert" | tail -5
