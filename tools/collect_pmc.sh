#!/bin/bash
# PMC evidence of the round, on the GPU box from the repo root (each rocprofv3 pass: --kernel-trace + --pmc only, the program directly after `--`):
#   gpurun_out/pmc5/attn_a, attn_b   tools/attn_bench.py 0           (attention kernels alone, cold operands rotated)
#   gpurun_out/pmc5/gemm             tools/gemm_bench.py 5 0         (the step's GEMM shapes with their epilogues)
#   gpurun_out/pmc5/step             bench.py --steps 2 --warmup 1   (the same counters for every kernel INSIDE the bench step)
# summarised by tools/pmc_summary.py into gpurun_out/pmc5/*.txt
export TMPDIR=/tmp
O=gpurun_out/pmc5
rm -rf $O; mkdir -p $O
A="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
B="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/attn_a -- python3 tools/attn_bench.py 0 > $O/attn_a.log 2>&1
rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/attn_b -- python3 tools/attn_bench.py 0 > $O/attn_b.log 2>&1
export GEMM_SHAPES=llm.wqkv,llm.wo+r,llm.w13,llm.w2+r,vit.qkv,vit.proj+r,vit.lin1,vit.lin2+r
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/gemm -- python3 tools/gemm_bench.py 5 0 > $O/gemm.log 2>&1
unset GEMM_SHAPES
export ULLSAM_BENCH_NO_GEMM_EVENTS=1
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/step -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-iou > $O/step.log 2>&1
for d in attn_a attn_b gemm step; do python3 tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1; done
grep "variant 0" $O/attn_a.log
tail -3 $O/gemm.log | cut -c1-200
find $O -name "*.csv" -delete; find $O -type d -empty -delete
wc -l $O/*.txt
