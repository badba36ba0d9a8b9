"""training.AttentionFn forward + backward at the step's three shapes (ViT-H windowed: 25 windows x 16 heads x 196 x 196 x 80; ViT-H global: 16 heads x 4096 x 4096 x 80;
InternLM2-7B: 32 heads / 8 KV heads x 1081 x 1081 x 128, causal), bf16 products, with a switch on and off (argv[1]: INPLACE_ATTN = the products in place on the row tensors vs head-major copies; matmul_vec = 16-byte operand fetches in the bf16 product)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ullsam_amd import training as T
DEV = "cuda:0"
KNOB = sys.argv[1] if len(sys.argv) > 1 else "INPLACE_ATTN"      # or FUSED_CAUSAL_FWD (the LLM forward on the inference causal kernel), or "matmul_vec": 16-byte fetches of the bf16 product's k-fastest operands (ullsam_train_set_matmul_vec)
def setk(on):
    if KNOB == "matmul_vec":
        from ullsam_amd import _lib
        _lib.load().ullsam_train_set_matmul_vec(int(on))
    else:
        setattr(T, KNOB, on)
g = torch.Generator(device=DEV); g.manual_seed(0)
SHAPES = [("vit windowed", 25, 16, 16, 196, 80, -1, 14), ("vit global", 1, 16, 16, 4096, 80, -1, 64), ("llm causal", 1, 32, 8, 1081, 128, 0, 0)]
for name, B, H, KVH, S, hd, causal, kw in SHAPES:
    q = torch.randn(B * S, H * hd, device=DEV, generator=g, requires_grad=True)
    k = torch.randn(B * S, KVH * hd, device=DEV, generator=g, requires_grad=True)
    v = torch.randn(B * S, KVH * hd, device=DEV, generator=g, requires_grad=True)
    bh = bw = None
    if kw:
        bh = torch.randn(B, H, S, kw, device=DEV, generator=g) * 0.1
        bw = torch.randn(B, H, S, kw, device=DEV, generator=g) * 0.1
    go = torch.randn(B * S, H * hd, device=DEV, generator=g)
    res = {}
    for mode in (False, True, False, True):
        setk(mode)
        ts = []
        for it in range(6):
            for t in (q, k, v): t.grad = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = T.AttentionFn.apply(q, k, v, B, H, KVH, S, S, causal, None, bh, bw, kw, True)
            out.backward(go)
            e1.record(); torch.cuda.synchronize()
            if it >= 2: ts.append(e0.elapsed_time(e1))
        res.setdefault(mode, []).extend(ts)
        last = (out.detach().clone(), q.grad.clone(), k.grad.clone(), v.grad.clone())
        res[("o", mode)] = last
    setk(True)
    m = {md: sorted(res[md])[len(res[md]) // 2] for md in (False, True)}
    d = max(float((a - b).abs().max()) for a, b in zip(res[("o", False)], res[("o", True)]))
    print(f"{name:13s}: {KNOB} off {m[False]:7.3f} ms   on {m[True]:7.3f} ms  ({100 * (m[True] / m[False] - 1):+.1f} %)   max |diff| over out / dq / dk / dv {d:.2e}", flush=True)
