"""A/B the GEMM kernel variants on the workload's shapes in ONE process (interleaved rounds, random data).
usage: python tools/gemm_bench.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ullsam_amd import ops, _lib

SHAPES = [  # (name, M, N, K, act)
    ("llm.wqkv", 4324, 6144, 4096, 0), ("llm.wo", 4324, 4096, 4096, 0), ("llm.w13", 4324, 28672, 4096, 3),
    ("llm.w2", 4324, 4096, 14336, 0), ("vitb.qkv", 16384, 2304, 768, 0), ("2b.wo", 4324, 2048, 2048, 0), ("mlp1.fc2", 4096, 4096, 4096, 0), ("vit.qkv", 16384, 3840, 1280, 0), ("vit.proj", 16384, 1280, 1280, 0),
    ("vit.lin1", 16384, 5120, 1280, 1), ("vit.lin2", 16384, 1280, 5120, 0), ("2b.w13", 4324, 16384, 2048, 3),
    ("vitb.lin1", 4096, 3072, 768, 1),
    # act code + 16: fp32 output with an fp32 residual updated in place (the residual-stream GEMMs of both transformers)
    ("dec.kproj", 262144, 128, 256, 0), ("dec.up1", 262144, 256, 256, 0),
    # the training step's frozen-LLM products at one 1081-token sequence (M = 1081: 9 x 128-row tiles x 32 = 288 workgroups of the 128x128 kernel = 1.125 rounds)
    ("trn.wo", 1081, 4096, 4096, 0), ("trn.w2", 1081, 4096, 14336, 0), ("trn.dw13", 1081, 4096, 28672, 0), ("trn.wqkv", 1081, 6144, 4096, 0), ("trn.dw2", 1081, 14336, 4096, 0),
    # the two shapes that trail the vendor library most, WITHOUT their epilogues (bias + GELU; SwiGLU pair): loop against loop
    ("vit.lin1.plain", 16384, 5120, 1280, 0), ("llm.w13.plain", 4324, 28672, 4096, 0),
    ("vit.proj+r", 16384, 1280, 1280, 16), ("vit.lin2+r", 16384, 1280, 5120, 16), ("llm.wo+r", 4324, 4096, 4096, 16), ("llm.w2+r", 4324, 4096, 14336, 16),
]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "3"]   # "9" = ullsam_set_gemm_variant(9); "0m3" = auto dispatch with ring-shape mask 3 (tuning key 1)

    def select(v):
        head, _, mk = str(v).partition("m")
        lib.ullsam_set_gemm_tuning(1, int(mk) if mk else 7)
        if int(head) >= 0:
            lib.ullsam_set_gemm_variant(int(head))
        return int(head)
    lib = _lib.load()
    if os.environ.get("GEMM_GM"):   # raster group height of the 256x256 kernels (A/B)
        lib.ullsam_set_gemm_tuning(0, int(os.environ["GEMM_GM"]))
    dev = "cuda"
    res = {}
    only = os.environ.get("GEMM_SHAPES")
    for name, M, N, K, act in SHAPES:
        if only and name not in only.split(","):
            continue
        res = act >= 16
        act &= 15
        # cold-operand regime: rotate through > 600 MB of distinct (A, W, C) so nothing is served from L2 / Infinity Cache
        n_out = N // 2 if act == 3 else N
        ncopy = max(2, int(6e8 // (2 * (M * K + N * K + M * n_out))) + 1)
        As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(ncopy)]
        Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(ncopy)]
        Cs = [torch.zeros(M, n_out, device=dev, dtype=torch.float32 if res else torch.bfloat16) for _ in range(ncopy)]
        a, w = As[0], Ws[0]
        bias = torch.randn(N, device=dev) if act == 1 else None
        ref = None
        skip = False
        for v in variants:
            if select(v) < 0:
                continue
            try:
                out = ops.gemm(a, w, bias, act=act, out_f32=res)
            except Exception as e:
                print(f"{name}: variant {v} cannot run this shape ({e})"); skip = True; break
            if ref is None:
                ref = out.float()
            else:
                d = (out.float() - ref).abs().max().item()
                assert d < 0.1 or select(v) > 15, (name, v, d)
        if skip:
            lib.ullsam_set_gemm_variant(0)
            continue
        times = {v: [] for v in variants}
        for r in range(rounds):
            for v in variants:
                vi = select(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(ncopy):
                    if vi < 0:   # comparator only: the vendor library's plain GEMM (no epilogue) through torch
                        torch.nn.functional.linear(As[i], Ws[i])
                    else:
                        ops.gemm(As[i], Ws[i], bias, act=act, out=Cs[i], out_f32=res, residual=Cs[i] if res else None)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / ncopy)
        fl = 2.0 * M * N * K
        line = f"{name:10s} M={M:6d} N={N:6d} K={K:6d}"
        for v in variants:
            t = sorted(times[v])[len(times[v]) // 2]
            line += f" | v{v}: {t * 1e3:8.1f} us {fl / t / 1e9:7.1f} TF/s"
        print(line, flush=True)
    lib.ullsam_set_gemm_variant(0)
    lib.ullsam_set_gemm_tuning(1, 7)


if __name__ == "__main__":
    main()
