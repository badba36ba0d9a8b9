set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "epilogue or wqkv" -s 2>&1 | tail -15 > gpurun_out/t1.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "full_depth" -s 2>&1 | tail -40 > gpurun_out/t2.log
python bench.py --steps 10 --warmup 3 > gpurun_out/bench1.log 2>&1
python tools/probes/amg_full_probe.py > gpurun_out/amg_probe.log 2>&1
tail -5 gpurun_out/t1.log; tail -25 gpurun_out/t2.log; tail -2 gpurun_out/bench1.log; tail -8 gpurun_out/amg_probe.log
