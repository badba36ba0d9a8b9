"""Greedy decode throughput of the InternLM2-7B-shaped LLM (app.py:431-495 caption path): prefill S tokens, then n new tokens.
usage: python tools/decode_bench.py [batch] [prompt_len] [new_tokens]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1081
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if os.environ.get("ULLSAM_GEMM_VARIANT"):
    from ullsam_amd import _lib
    _lib.load().ullsam_set_gemm_variant(int(os.environ["ULLSAM_GEMM_VARIANT"]))
m = build_model("b", "7b", torch.bfloat16, "cuda:0")
lm = m.language_model
ids = torch.randint(0, 90000, (B, S), device="cuda")
def run(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = lm.generate(input_ids=ids, max_new_tokens=k, eos_token_id=-1)
    torch.cuda.synchronize(); return time.perf_counter() - t0, out
run(4)
t1, _ = run(1)
tn, out = run(n)
per = (tn - t1) / (n - 1)
print(json.dumps({"workload": f"greedy decode, InternLM2-7B-shaped, bf16, batch {B}, prompt {S}", "prefill_plus_first_token_ms": round(t1 * 1e3, 1),
                  "ms_per_decode_step": round(per * 1e3, 3), "tokens_per_s": round(B / per, 1),
                  "weight_bytes_per_step_GB": 15.5, "hbm_floor_ms": round(15.5e9 / 5e12 * 1e3, 2)}))
