"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals (short names) and per-grid GEMM groups.
usage: python tools/prof_summary.py <kernel_trace.csv> [steps_in_trace]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I", n)
    if n.startswith("_Z"):
        base = re.match(r"_Z(\d+)", n)
        k = int(base.group(1))
        name = n[2 + len(base.group(1)):][:k]
        rest = n[2 + len(base.group(1)) + k:]
        tag = ("bf16" if "DF16b" in rest else "f32") + "".join(re.findall(r"Li(\d+)E", rest) and ["<" + ",".join(re.findall(r"Li(\d+)E", rest)) + ">"] or [])
        return f"{name}[{tag}]"
    return n.split("(")[0][:70]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = list(csv.DictReader(open(path)))
    tot = defaultdict(lambda: [0, 0.0])
    gemm = defaultdict(lambda: [0, 0.0])
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        n = short(r["Kernel_Name"])
        tot[n][0] += 1
        tot[n][1] += d
        if "gemm" in n:
            g = (n, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
            gemm[g][0] += 1
            gemm[g][1] += d
    allus = sum(v[1] for v in tot.values())
    print(f"# {path}: {len(rows)} dispatches, {allus / 1e3:.1f} ms GPU time ({allus / 1e3 / steps:.1f} ms per step over {steps:g} steps)")
    print(f"{'kernel':58s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s}")
    for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{n[:58]:58s} {c:6d} {us / 1e3:9.2f} {us / c:9.1f} {100 * us / allus:5.1f}%")
    print("\n# GEMM launches grouped by workgroup count")
    for (n, g), (c, us) in sorted(gemm.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"{n[:34]:34s} wgs={g:6d} calls={c:5d} total_ms={us / 1e3:8.2f} avg_us={us / c:8.1f}")


if __name__ == "__main__":
    main()
