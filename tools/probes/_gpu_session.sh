cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
export TMPDIR=/tmp
SHA=34f06dc
timeout 900 python3 tools/collect_evidence.py --round 03 --head $SHA --mode mask > gpurun_out/ev/mask.log 2>&1 < /dev/null
timeout 600 python3 tools/collect_evidence.py --round 03 --head $SHA --mode decode > gpurun_out/ev/decode.log 2>&1 < /dev/null
cp profiles/r03_kernel_summary_HEAD.txt profiles/r03_pmc_bench_traffic.json profiles/r03_decode_summary.txt gpurun_out/ev/
timeout 600 python3 bench.py > gpurun_out/ev/bench_mask.json 2> gpurun_out/ev/bench_mask.err < /dev/null
python3 -c "
import json; d=json.loads(open('gpurun_out/ev/bench_mask.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['mask_iou_vs_fp32']['mean'])"
