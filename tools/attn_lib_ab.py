"""Same-process A/B of the three attention launches of the bench step (ViT-H windowed / global at batch 4, InternLM2-7B causal prefill) under several
BUILDS of the kernel library (tools/build_side.py / side builds named libullsam_hip_<name>.so; "cur" = the committed build), interleaved rounds.
usage: python tools/attn_lib_ab.py [rounds] nameA,nameB,...   (a name may carry an attention variant: "cur:12")"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ullsam_amd import ops, _lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["cur:12", "cur"]
libs = {}
for n in names:
    base = n.split(":")[0]
    if base not in libs:
        _lib._lib = None
        _lib.LIB_PATH = os.path.join(ROOT, "ullsam_amd", "lib", "libullsam_hip.so" if base == "cur" else f"libullsam_hip_{base}.so")
        libs[base] = _lib.load()
dev = "cuda"
B, heads, hd = 4, 16, 80
qkv = [torch.randn(B * 4096, 3 * heads * hd, device=dev).bfloat16() for _ in range(4)]
bias = torch.randn(3 * heads * hd, device=dev).bfloat16()
rh14, rw14 = (torch.randn(27, hd, device=dev) * 0.1).bfloat16(), (torch.randn(27, hd, device=dev) * 0.1).bfloat16()
rh64, rw64 = (torch.randn(127, hd, device=dev) * 0.1).bfloat16(), (torch.randn(127, hd, device=dev) * 0.1).bfloat16()
H, KVH, S = 32, 8, 1081
q = torch.randn(B * S, H * 128, device=dev).bfloat16()
kc = torch.randn(B, KVH, S, 128, device=dev).bfloat16(); vc = torch.randn(B, KVH, S, 128, device=dev).bfloat16()
cases = {"window": lambda i: ops.vit_attention(qkv[i], rh14, rw14, bias, B, heads, hd, 64, 64, 14),
         "global": lambda i: ops.vit_attention(qkv[i], rh64, rw64, bias, B, heads, hd, 64, 64, 0),
         "causal": lambda i: ops.causal_attention(q, kc, vc, None, B, H, KVH, 128, S, S, 0)}
only = os.environ.get("ATTN_CASES")
if only:
    cases = {k: v for k, v in cases.items() if k in only.split(",")}


def select(n):
    base, _, var = n.partition(":")
    _lib._lib = libs[base]
    libs[base].ullsam_set_attn_variant(int(var) if var else 0)


times = {(n, c): [] for n in names for c in cases}
for n in names:
    select(n)
    for c, fn in cases.items():
        for i in range(4):
            fn(i)
torch.cuda.synchronize()
for r in range(rounds):
    for n in (names if r % 2 == 0 else names[::-1]):
        select(n)
        for c, fn in cases.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(4):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            times[(n, c)].append(e0.elapsed_time(e1) / 4 * 1e3)
for n in names:
    select(n)
    libs[n.split(":")[0]].ullsam_set_attn_variant(0)
    print(f"{n:10s} " + "   ".join(f"{c} {sorted(times[(n, c)])[rounds // 2]:7.1f} us (min {min(times[(n, c)]):6.1f})" for c in cases), flush=True)
