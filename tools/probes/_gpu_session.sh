cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "decode or generate or gemm" 2>&1 < /dev/null | tail -4
timeout 300 python bench.py --mode decode --no-cpu-baseline 2>&1 < /dev/null | tail -1 > gpurun_out/dec.log
python3 -c "
import json; d=json.loads(open('gpurun_out/dec.log').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
