cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 < /dev/null | tail -8 > gpurun_out/full_gpu.log
cat gpurun_out/full_gpu.log
