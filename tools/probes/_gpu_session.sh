cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "causal" 2>&1 | tail -4 > gpurun_out/t1.log
python tools/attn_bench.py 11,0 2>&1 | grep causal >> gpurun_out/t1.log
python tools/probes/causal_stamps.py >> gpurun_out/t1.log 2>&1
cat gpurun_out/t1.log
