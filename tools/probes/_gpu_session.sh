cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_all.log
GEMM_SHAPES=llm.wqkv,llm.wo,vitb.qkv,2b.wo,mlp1.fc2,vit.qkv,vitb.lin1,2b.w13 python tools/gemm_bench.py 5 3,6 > gpurun_out/gb.log 2>&1
python tools/step_ab.py 6 6,7 > gpurun_out/ab.log 2>&1
tail -8 gpurun_out/t_all.log; cat gpurun_out/gb.log | tail -9; tail -3 gpurun_out/ab.log
