#!/bin/bash
# rocprofv3 kernel trace of tools/train_step_bench.py h 7b 2 bf16 (3 steps) -> gpurun_out/train_summary.txt.  Run on the GPU box from the repo root.
export TMPDIR=/tmp
rm -rf gpurun_out/trainprof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trainprof -o run -- python3 tools/train_step_bench.py h 7b 2 bf16 > gpurun_out/trainprof.log 2>&1
f=$(find gpurun_out/trainprof -name "*kernel_trace.csv" | head -1)
python3 tools/prof_summary.py "$f" 3 > gpurun_out/train_summary.txt
grep forward_s gpurun_out/trainprof.log | cut -c1-300
head -${1:-40} gpurun_out/train_summary.txt
rm -rf gpurun_out/trainprof
