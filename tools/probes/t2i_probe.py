import math, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from ullsam_amd import ops, _lib
L = _lib.load()
def sm(x):
    x = x - x.max(-1, keepdims=True); e = np.exp(x); return e / e.sum(-1, keepdims=True)
for (P, Tq, N) in [(1, 7, 4096), (3, 8, 1000), (70, 6, 4096), (2, 1, 64), (5, 7, 130), (4, 12, 777), (64, 16, 4096), (2, 9, 33)]:
    rng = np.random.default_rng(P * 1000 + Tq)
    H, hd = 8, 16
    q = rng.standard_normal((P, Tq, 128), dtype=np.float32) * 2
    k = rng.standard_normal((P, N, 128), dtype=np.float32); v = rng.standard_normal((P, N, 128), dtype=np.float32)
    kd, vd = torch.from_numpy(k).cuda().bfloat16(), torch.from_numpy(v).cuda().bfloat16()
    kr, vr = kd.float().cpu().numpy().astype(np.float64), vd.float().cpu().numpy().astype(np.float64)
    sp = lambda x: x.reshape(x.shape[0], x.shape[1], H, hd).transpose(0, 2, 1, 3)
    a = sm(np.matmul(sp(q.astype(np.float64)), sp(kr).transpose(0, 1, 3, 2)) / math.sqrt(hd))
    ref = np.matmul(a, sp(vr)).transpose(0, 2, 1, 3).reshape(q.shape)
    res = {}
    for var in ([0, 16] if Tq <= 8 else [0]):
        L.ullsam_set_attn_variant(var)
        out = ops.tok2img_attention(torch.from_numpy(q).cuda().reshape(P * Tq, -1), kd.reshape(P * N, -1), vd.reshape(P * N, -1), P, H, hd, Tq, N, 1 / math.sqrt(hd))
        torch.cuda.synchronize()
        o = out.cpu().numpy().reshape(q.shape)
        res[var] = float(np.abs(o - ref).max() / np.abs(ref).max())
        # timing
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        qq = torch.from_numpy(q).cuda().reshape(P * Tq, -1); kk = kd.reshape(P * N, -1); vv = vd.reshape(P * N, -1)
        for _ in range(3): ops.tok2img_attention(qq, kk, vv, P, H, hd, Tq, N, 1 / math.sqrt(hd))
        e0.record()
        for _ in range(20): ops.tok2img_attention(qq, kk, vv, P, H, hd, Tq, N, 1 / math.sqrt(hd))
        e1.record(); torch.cuda.synchronize()
        res[f"us{var}"] = round(e0.elapsed_time(e1) * 50, 1)
    L.ullsam_set_attn_variant(0)
    print(P, Tq, N, res, flush=True)
