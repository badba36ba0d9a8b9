#!/bin/bash
# Which kernels the vendor library (hipBLASLt / rocBLAS through torch.nn.functional.linear) runs on the step's GEMM shapes: rocprofv3 --kernel-trace over the comparator-only
# variant of tools/gemm_bench.py -> gpurun_out/vendor_names.txt (kernel name, calls, mean us per shape; the name encodes macro tile MT, depth, LDS / direct-to-LDS flags).
export TMPDIR=/tmp
O=gpurun_out/vendor_trace
rm -rf $O; mkdir -p $O
: > gpurun_out/vendor_names.txt
for sh in vit.lin1.plain llm.w13.plain vit.qkv llm.wqkv; do
  export GEMM_SHAPES=$sh
  rocprofv3 --kernel-trace --output-format csv -d $O/$sh -- python3 tools/gemm_bench.py 2 -1 > $O/$sh.log 2>&1
  f=$(find $O/$sh -name "*kernel_trace.csv" | head -1)
  echo "## $sh" >> gpurun_out/vendor_names.txt
  python3 - "$f" >> gpurun_out/vendor_names.txt <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0, None])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "Cijk" in n or "gemm" in n.lower() and "ullsam" not in n:
        a = agg[n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[2] = (r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""))
for n, (c, us, meta) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:3]:
    print(f"calls={c} mean_us={us / c:.1f} grid={meta[0]} wg={meta[1]} lds={meta[2]} vgpr={meta[3]} agpr={meta[4]} sgpr={meta[5]}\n  {n}")
PY
done
rm -rf $O
cat gpurun_out/vendor_names.txt
