#!/bin/bash
# PMC passes over the automatic mask generator (2 tiles): instruction / matrix-pipe counters, then FETCH_SIZE and WRITE_SIZE in passes of their own -> gpurun_out/pmc5amg/*.txt
export TMPDIR=/tmp
O=gpurun_out/pmc5amg
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $O/a -- python3 tools/amg_bench.py 64 2048 h 0.92 1.0 2 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 tools/amg_bench.py 64 2048 h 0.92 1.0 2 > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 tools/amg_bench.py 64 2048 h 0.92 1.0 2 > $O/w.log 2>&1
export PMC_FILTER="i2t_block|kv_proj|amg_postprocess|dec_tok|dec_heads|tok2img|up1_ln|up2_hyper|rle_emit|gemm128|gemm_ring8" PMC_TOP=20
for d in a f w; do python3 tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1; done
find $O -name "*.csv" -delete; find $O -type d -empty -delete
cut -c1-260 $O/a.txt | head -14; cut -c1-160 $O/f.txt | head -14; cut -c1-160 $O/w.txt | head -14
