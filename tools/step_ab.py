"""Same-process A/B of the bench step (bench.py's model, inputs and step) under GEMM dispatch variants: boxes differ by up to 12 % and
drift with load, so only interleaved rounds in one process compare two dispatch policies.
usage: python tools/step_ab.py [rounds] [maskA,maskB,...]     ring tile shapes the auto dispatch may pick (ullsam_set_gemm_tuning key 1):
0 two-buffer / 128x128 kernels only, bit 0 256x256 ring, bit 1 256x320 ring, bit 2 272x256 ring (default 7)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ullsam_amd import _lib


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "6", "7"]
    lib = _lib.load()
    dev = "cuda"
    model = bench.build_model("h", "7b", torch.bfloat16, dev)
    inputs = bench.make_inputs(4, 1081, dev, True)
    step = bench.mask_path_compute(model, inputs, torch.bfloat16)
    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        times = {v: [] for v in variants}
        for r in range(rounds):
            for v in (variants if r % 2 == 0 else variants[::-1]):   # ABBA: the variant measured second in a round comes out ~0.5 % faster
                head, _, pz = str(v).partition("p")     # "7p0" = dispatch mask 7 with the one-tile ring kernel (ullsam_set_gemm_tuning key 2: 0 one tile per workgroup, 2 persistent = the default, 1 / 4 the earlier persistent forms)
                lib.ullsam_set_gemm_tuning(2, int(pz) if pz else 2)   # (2 = the library's default)
                head, _, gm = head.partition("g")       # "7g8" = dispatch mask 7 with raster groups of 8 tile rows
                lib.ullsam_set_gemm_tuning(0, int(gm) if gm else 4)
                head, _, av = head.partition("a")        # "7a9" = dispatch mask 7 with ullsam_set_attn_variant(9)
                mask, _, gv = head.partition("v")        # "7v64" = dispatch mask 7 with ullsam_set_gemm_variant(64) (no split-K tails)
                lib.ullsam_set_gemm_tuning(1, int(mask))
                lib.ullsam_set_gemm_variant(int(gv) if gv else 0)
                lib.ullsam_set_attn_variant(int(av) if av else 0)
                step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    step()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 4)
    lib.ullsam_set_gemm_tuning(1, 7)
    lib.ullsam_set_gemm_variant(0)
    lib.ullsam_set_attn_variant(0)
    lib.ullsam_set_gemm_tuning(0, 4)
    lib.ullsam_set_gemm_tuning(2, 2)
    for v in variants:
        t = sorted(times[v])
        print(f"dispatch mask {v}: median {t[len(t) // 2]:.3f} ms/step  (min {t[0]:.3f}, max {t[-1]:.3f})  = {4e3 / t[len(t) // 2]:.2f} images/s")


if __name__ == "__main__":
    main()
