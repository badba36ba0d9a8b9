cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "causal" 2>&1 | tail -6 > gpurun_out/t1.log
python tools/attn_bench.py 11,0 > gpurun_out/attn.log 2>&1
tail -4 gpurun_out/t1.log; cat gpurun_out/attn.log | tail -5
