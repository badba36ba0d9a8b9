cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ring or 320 or 272 or epilogue or wqkv or persistent" -s 2>&1 | tail -12 > gpurun_out/t1.log
GEMM_SHAPES=llm.wo+r,llm.w13,llm.w2+r,vit.qkv,vit.proj+r,vit.lin1,vit.lin2+r python tools/gemm_bench.py 5 0L0,0L1 > gpurun_out/gb.log 2>&1
python tools/step_ab.py 6 31L0,31L1 > gpurun_out/ab.log 2>&1
python -m pytest tests/test_amg_gpu.py -x -q -m gpu -k "real_size" -s 2>&1 | tail -12 > gpurun_out/t3.log
tail -6 gpurun_out/t1.log; cat gpurun_out/gb.log | tail -8; tail -3 gpurun_out/ab.log; tail -8 gpurun_out/t3.log
