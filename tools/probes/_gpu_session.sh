cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "decode or generate or skinny or llm" 2>&1 | tail -4 > gpurun_out/t1.log
python bench.py --mode decode --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/dec.log
cat gpurun_out/t1.log; cut -c1-330 gpurun_out/dec.log
