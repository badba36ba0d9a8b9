cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "sam_forward" 2>&1 | tail -12 > gpurun_out/t1.log
python bench.py --llm none --batch 8 --no-cpu-baseline > gpurun_out/b16.log 2>&1
tail -5 gpurun_out/t1.log; tail -1 gpurun_out/b16.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['gemm_share_of_step'], d['mask_iou_vs_fp32']['mean'])"
