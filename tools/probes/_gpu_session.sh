cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "causal or llm or ullsam_tiny or full_depth or generate or greedy or batched" 2>&1 | tail -4 > gpurun_out/t1.log
python tools/attn_bench.py 0 2>&1 | grep causal >> gpurun_out/t1.log
python tools/step_ab.py 6 7a11,7 > gpurun_out/ab.log 2>&1
cat gpurun_out/t1.log; tail -2 gpurun_out/ab.log
