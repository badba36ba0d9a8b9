cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
H=$(cat gpurun_out/HEAD 2>/dev/null || echo unknown)
python tools/collect_evidence.py --round 03 --head $H > gpurun_out/evidence.log 2>&1
python tools/collect_evidence.py --round 03 --head $H --mode decode > gpurun_out/evidence_decode.log 2>&1
python tools/other_configs.py --round 03 --head $H > gpurun_out/other.log 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc_attn_a -- python3 tools/attn_bench.py 0 > gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_attn_b -- python3 tools/attn_bench.py 0 > gpurun_out/pmc_b.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_attn_a gpurun_out/pmc_attn_b > gpurun_out/r03_pmc_attention_raw.txt 2>&1
GEMM_SHAPES=llm.wqkv,llm.wo+r,llm.w13,llm.w2+r,vit.qkv,vit.proj+r,vit.lin1,vit.lin2+r rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pmc_gemm -- python3 tools/gemm_bench.py 2 0 > gpurun_out/pmc_g.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_gemm > gpurun_out/r03_pmc_gemm_raw.txt 2>&1
python tools/probes/ring8_stamps.py w13 wo w2 qkv lin2 > gpurun_out/r03_ring_stamps.txt 2>&1
python tools/probes/causal_stamps.py > gpurun_out/r03_causal_stamps.txt 2>&1
python bench.py > gpurun_out/bench_default.log 2>&1
cp profiles/r03_* gpurun_out/ 2>/dev/null
tail -3 gpurun_out/evidence.log | cut -c1-300; tail -2 gpurun_out/other.log | cut -c1-300; tail -1 gpurun_out/bench_default.log | cut -c1-1500
