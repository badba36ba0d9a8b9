"""Join rocprofv3 counter_collection CSVs with kernel traces: per (kernel, grid) average counter values + duration.
usage: [PMC_FILTER=<regex of kernel names>] python tools/pmc_summary.py <dir> [<dir> ...]   (each dir = one --pmc pass; default filter: gemm|attn|norm)"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I", n)
    base = re.match(r"_Z(\d+)", n)
    if base:
        k = int(base.group(1)); name = n[2 + len(base.group(1)):][:k]
        return name + ("[bf16]" if "DF16b" in n else "[f32]")
    return n.split("(")[0][:50]


FILTER = re.compile(os.environ.get("PMC_FILTER", "gemm|attn|norm"))


def main():
    for d in sys.argv[1:]:
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
        dur = {}
        for r in csv.DictReader(open(kt)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int); seen = set()
        for r in csv.DictReader(open(cc)):
            name = short(r["Kernel_Name"])
            if not FILTER.search(name):
                continue
            key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Dispatch_Id"], key) not in seen:
                seen.add((r["Dispatch_Id"], key)); cnt[key] += 1
                agg[key]["_us"] += dur.get(r["Dispatch_Id"], 0.0)
        print(f"# {d}")
        for key, cs in sorted(agg.items(), key=lambda kv: -kv[1]["_us"])[:int(os.environ.get("PMC_TOP", "12"))]:
            n = cnt[key]
            print(f"{key[0]:28s} wgs={key[1]:6d} n={n:4d} avg_us={cs['_us'] / n:9.1f} " + " ".join(f"{k}={v / n:.4g}" for k, v in sorted(cs.items()) if k != "_us"))


if __name__ == "__main__":
    main()
