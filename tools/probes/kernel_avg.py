"""Average duration of the kernels whose name contains a pattern, from a rocprofv3 kernel_trace.csv.
usage: python tools/probes/kernel_avg.py <dir> <pattern>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if sys.argv[2] in n:
        d.setdefault(n[:90], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    v.sort()
    print(f"{n}: {len(v)} calls, median {v[len(v) // 2]:.1f} us, mean {sum(v) / len(v):.1f} us")
