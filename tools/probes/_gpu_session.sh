cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "fused_launches" 2>&1 < /dev/null | tail -12 > gpurun_out/t1.log
cat gpurun_out/t1.log
timeout 300 python bench.py --mode decode --no-cpu-baseline 2>&1 < /dev/null | tail -1 > gpurun_out/dec.log
python3 -c "
import json; d=json.loads(open('gpurun_out/dec.log').read()); print(d['value'], d['ms_per_step'], d['roofline'])"
