"""Median GPU-side duration per (kernel, grid) from a rocprofv3 --kernel-trace csv.  usage: ktrace_summary.py DIR [substring]"""
import collections, csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
sub = sys.argv[2] if len(sys.argv) > 2 else ''
d = collections.defaultdict(list)
rows = list(csv.DictReader(open(fs[0])))
for r in rows:
    if sub in r['Kernel_Name']:
        d[(r['Kernel_Name'][:48], r.get('Grid_Size_X', r.get('Grid_Size', '')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:50s} grid {k[1]:>8s} n={len(v):5d} median {v[len(v) // 2]:8.2f} us  min {v[0]:8.2f}  total {sum(v) / 1e3:8.2f} ms")
