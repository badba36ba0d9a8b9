#!/bin/bash
# Matrix-pipe busy, shader clock and LDS instruction counts of the VENDOR library's GEMM kernel next to this library's, same shapes, same process:
# rocprofv3 --kernel-trace --pmc over tools/gemm_bench.py <rounds> 0,-1 per shape -> gpurun_out/vendor_pmc.txt
export TMPDIR=/tmp
O=gpurun_out/vendor_pmc
rm -rf $O; mkdir -p $O
A="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
B="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
: > gpurun_out/vendor_pmc.txt
for sh in ${VENDOR_PMC_SHAPES:-llm.w13.plain vit.lin1.plain vit.qkv llm.wqkv}; do
  export GEMM_SHAPES=$sh
  rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/a_$sh -- python3 tools/gemm_bench.py 3 0,-1 > $O/a_$sh.log 2>&1
  rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/b_$sh -- python3 tools/gemm_bench.py 3 0,-1 > $O/b_$sh.log 2>&1
  echo "## $sh" >> gpurun_out/vendor_pmc.txt
  PMC_FILTER="gemm|Cijk" PMC_TOP=4 python3 tools/pmc_summary.py $O/a_$sh $O/b_$sh >> gpurun_out/vendor_pmc.txt 2>&1
  tail -1 $O/a_$sh.log | cut -c1-200 >> gpurun_out/vendor_pmc.txt
done
find $O -name "*.csv" -delete; find $O -type d -empty -delete
cat gpurun_out/vendor_pmc.txt
