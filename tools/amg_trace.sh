#!/bin/bash
# rocprofv3 kernel trace of tools/amg_bench.py over 6 tiles -> gpurun_out/amg_summary.txt (per-kernel table; tools/prof_summary.py).  Run on the GPU box from the repo root.
export TMPDIR=/tmp
rm -rf gpurun_out/amgprof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/amgprof -o run -- python3 tools/amg_bench.py 64 2048 h 0.92 1.0 6 > gpurun_out/amgprof.log 2>&1
f=$(find gpurun_out/amgprof -name "*kernel_trace.csv" | head -1)
python3 tools/prof_summary.py "$f" 6 > gpurun_out/amg_summary.txt
grep seconds_per_tile gpurun_out/amgprof.log | cut -c1-200
head -${1:-24} gpurun_out/amg_summary.txt
rm -rf gpurun_out/amgprof
