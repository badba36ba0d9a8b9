"""Same-process A/B of the bench step under two BUILDS of the kernel library (the committed one vs a side build made by
tools/build_side.py): boxes differ by up to 12 % and drift with load, so a kernel change is judged by interleaved rounds in one process.

    python tools/lib_ab.py [rounds] nameA,nameB[,...]      name = "cur" (ullsam_amd/lib/libullsam_hip.so) or a side build's <name>
    LIB_AB_KERNELS=1: also print the per-GEMM-class event times (HIP events around every ullsam_gemm launch) per library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ullsam_amd import _lib


def load_libs(names):
    libs = {}
    for n in names:
        path = os.path.join(ROOT, "ullsam_amd", "lib", "libullsam_hip.so" if n == "cur" else f"libullsam_hip_{n}.so")
        _lib._lib = None
        _lib.LIB_PATH = path
        libs[n] = _lib.load()
    return libs


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["r03", "cur"]
    libs = load_libs(names)
    dev = "cuda"
    model = bench.build_model("h", "7b", torch.bfloat16, dev)
    inputs = bench.make_inputs(4, 1081, dev, True)
    step = bench.mask_path_compute(model, inputs, torch.bfloat16)
    outs = {}
    with torch.no_grad():
        for n in names:
            _lib._lib = libs[n]
            for _ in range(2):
                low, mk = step()
            outs[n] = (low.float().clone(), mk.clone())
        torch.cuda.synchronize()
        times = {n: [] for n in names}
        for r in range(rounds):
            for n in (names if r % 2 == 0 else names[::-1]):
                _lib._lib = libs[n]
                step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    step()
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / 4)
    ref = outs[names[0]]
    for n in names:
        t = sorted(times[n])
        d = (outs[n][0] - ref[0]).abs().max().item()
        flips = (outs[n][1] != ref[1]).float().mean().item()
        print(f"lib {n:8s}: median {t[len(t) // 2]:.3f} ms/step  (min {t[0]:.3f}, max {t[-1]:.3f})  = {4e3 / t[len(t) // 2]:.2f} images/s   "
              f"low-res logits vs {names[0]}: max abs diff {d:.3e}, mask pixels flipped {flips:.2e}", flush=True)
    if os.environ.get("LIB_AB_KERNELS"):
        from collections import defaultdict
        from ullsam_amd import ops
        orig, orig_rope = ops.gemm, ops.gemm_qkv_rope
        for n in names:
            _lib._lib = libs[n]
            rec = []

            def timed(a, w, *args, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); out = orig(a, w, *args, **kw); e1.record()
                rec.append(((a.shape[0], w.shape[0], a.shape[1], kw.get("act", 0), "res" if kw.get("residual") is not None else ""), e0, e1))
                return out

            def timed_rope(x, w, *args, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); out = orig_rope(x, w, *args, **kw); e1.record()
                rec.append(((x.shape[0], w.shape[0], x.shape[1], 4, "rope"), e0, e1))
                return out

            ops.gemm, ops.gemm_qkv_rope = timed, timed_rope
            with torch.no_grad():
                for _ in range(3):
                    step()
            torch.cuda.synchronize()
            ops.gemm, ops.gemm_qkv_rope = orig, orig_rope
            agg = defaultdict(list)
            for k, e0, e1 in rec:
                agg[k].append(e0.elapsed_time(e1) * 1e3)
            tot = sum(sum(v) for v in agg.values()) / 3
            print(f"--- lib {n}: GEMM classes (M, N, K, act, kind): launches/step, median us, TFLOP/s; total {tot / 1e3:.2f} ms/step")
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                if sum(v) / 3 < 100:
                    continue
                med = sorted(v)[len(v) // 2]
                print(f"    {str(k):44s} x{len(v) // 3:3d}  {med:8.1f} us  {2.0 * k[0] * k[1] * k[2] / med / 1e6:7.1f} TF/s   {sum(v) / 3 / 1e3:6.2f} ms/step")
    _lib._lib = libs[names[-1]]


if __name__ == "__main__":
    main()
