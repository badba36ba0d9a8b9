cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
export TMPDIR=/tmp
SHA=7705191
timeout 900 python3 tools/collect_evidence.py --round 03 --head $SHA --mode mask > gpurun_out/ev/mask.log 2>&1 < /dev/null
timeout 600 python3 tools/collect_evidence.py --round 03 --head $SHA --mode decode > gpurun_out/ev/decode.log 2>&1 < /dev/null
timeout 900 python3 tools/other_configs.py --round 03 --head $SHA > gpurun_out/ev/other.log 2>&1 < /dev/null
cp profiles/r03_kernel_summary_HEAD.txt profiles/r03_pmc_bench_traffic.json profiles/r03_decode_summary.txt profiles/r03_other_configs.json gpurun_out/ev/
timeout 600 python3 bench.py > gpurun_out/ev/bench_mask.json 2> gpurun_out/ev/bench_mask.err < /dev/null
timeout 300 python3 bench.py --mode decode > gpurun_out/ev/bench_decode.json 2> gpurun_out/ev/bench_decode.err < /dev/null
tail -3 gpurun_out/ev/mask.log; tail -2 gpurun_out/ev/decode.log; tail -3 gpurun_out/ev/other.log
tail -1 gpurun_out/ev/bench_mask.json | cut -c1-900
tail -1 gpurun_out/ev/bench_decode.json | cut -c1-300
