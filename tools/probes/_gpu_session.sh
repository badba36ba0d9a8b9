cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --llm none --batch 8 --no-cpu-baseline > gpurun_out/b16.log 2>&1
python bench.py --llm none --batch 8 --no-cpu-baseline --vit-fp8 > gpurun_out/b8.log 2>&1
tail -1 gpurun_out/b16.log | cut -c1-1800; echo; tail -1 gpurun_out/b8.log | cut -c1-1800
