cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "decode or generate" 2>&1 < /dev/null | tail -4
timeout 400 python3 tools/decode_bench.py 4 1081 64 2>&1 < /dev/null | tail -1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof2 -o sk -- python3 tools/decode_bench.py 4 1081 64 > gpurun_out/dec.log 2>&1 < /dev/null
timeout 120 python3 tools/probes/ktrace_summary.py /tmp/prof2 decode_attn < /dev/null | head -6
