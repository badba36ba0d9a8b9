for gm in 4 8 17 2; do echo "GM=$gm"; GEMM_GM=$gm GEMM_SHAPES=llm.w13,llm.wqkv,llm.w2,vit.qkv,vit.lin1 timeout 300 python tools/gemm_bench.py 5 128,7; done
