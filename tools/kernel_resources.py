"""Per-kernel register / scratch metadata of the gfx950 code objects inside libullsam_hip.so (what `hipcc -Rpass-analysis=kernel-resource-usage`
prints at build time, read back from the SHIPPED library).

    python tools/kernel_resources.py [--hot] [path/to/lib.so]

The library's `.hip_fatbin` section holds one clang offload bundle per translation unit; each bundle's `hipv4-amdgcn-amd-amdhsa--gfx950` entry is an
ELF code object whose NT_AMDGPU_METADATA note lists, per kernel, `.vgpr_count`, `.agpr_count`, `.sgpr_count`, `.vgpr_spill_count`,
`.sgpr_spill_count`, `.private_segment_fixed_size` (scratch bytes per lane) and `.group_segment_fixed_size` (static LDS).
tests/test_host_cpu.py::test_hot_kernels_compile_without_spills fails the build when a kernel on the bench step's path reports a spill.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ULLSAM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "ullsam_amd", "lib", "libullsam_hip.so")

# kernels the bench step (configs[2], bf16) and the decode step spend their time in: these must compile spill-free.  Mangled-name
# fragments: DF16b = __bf16, Li<N>E = an integer template argument, Lb<0|1>E = a bool.
HOT_BF16 = [r"gemm_ring8_kernelILi\d+ELi\d+ELi\d+ELb0E", r"gemm256_kernelIDF16bLi0E", r"gemm128_kernelIDF16b", r"flash_attn_kernelIDF16bLi80ELi2ELi8ELb1E",
            r"vitglob_attn_kernel", r"causal128_attn_kernel", r"win14_attn_kernel", r"win14r_attn_kernelILi0E", r"norm_kernelI\w*DF16b", r"norm_block_kernel", r"gemm_skinny",
            r"decode_attn"]
# round 5: the mask decoder's and the automatic mask generator's kernels (bf16 path): the fused two-way-block kernels, the upscaling kernels, the post-processing
# (gemm256_kernel<*, 1> -- the two-buffer kernel's LDS-staged RoPE epilogue, 16 - 28 spills -- is still dispatched, by fp32 wqkv and by bf16 operands the ring's
# epilogue cannot take (unaligned q / k / v): not on the bench path, not in this list)
HOT_AMG = [r"i2t_block_kernel", r"kv_proj_kernel", r"up1_ln_gelu_kernel", r"up2_hyper_kernel", r"tok2img_partial_mfma_kernel", r"tok2img_merge_kernel", r"dec_tok_attn_kernel", r"dec_tok_mlp_kernel",
           r"dec_heads_kernel", r"amg_postprocess_kernel", r"rle_emit_kernel"]
HOT_BF16 = HOT_BF16 + HOT_AMG
HOT = HOT_BF16
# round 6: the persistent ring kernel in its default schedule (SCHED 2, no stamps, no ablation): no spilled VECTOR register and no scratch.  Its tile loop keeps more scalars than the 102 SGPRs hold
# (tile coordinates, two buffer descriptors, the kernel arguments the epilogue reads): hipcc parks the overflow in lanes of a VGPR (v_writelane / v_readlane, outside the K loop) -- counted as
# sgpr_spill_count, no memory traffic -- so that count is reported, not gated.
HOT_PERSISTENT = [r"gemm_ring8p_kernelILi\d+ELi\d+ELi\d+ELi\d+ELb0ELi2ELi2ELi0E"]


def code_objects(lib: str):
    """-> list of gfx950 ELF images (bytes) bundled in the library."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernels(lib: str = DEFAULT_LIB):
    """-> {demangled kernel name: {field: int}}"""
    res = {}
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            rec = {k: int(v) for k, v in re.findall(r"\.(agpr_count|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", blk)}
            res[name.group(1)] = rec
    return res   # MANGLED names (binutils' c++filt garbles the DF16b = __bf16 template arguments)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ks = kernels(args[0] if args else DEFAULT_LIB)
    hot = "--hot" in sys.argv
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'scratch':>7} {'lds':>7}  kernel")
    for name in sorted(ks):
        if hot and not any(re.search(h, name) for h in HOT):
            continue
        r = ks[name]
        print(f"{r.get('vgpr_count', 0):5d} {r.get('agpr_count', 0):5d} {r.get('sgpr_count', 0):5d} {r.get('vgpr_spill_count', 0):6d} "
              f"{r.get('private_segment_fixed_size', 0):7d} {r.get('group_segment_fixed_size', 0):7d}  {name[:150]}")


if __name__ == "__main__":
    main()
