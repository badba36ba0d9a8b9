"""rocprofv3 --pmc csv -> one line per kernel name: mean of every counter (counter_collection.csv under the given directory)."""
import sys, csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in rows.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k, " ".join(f"{n}={sum(v) / len(v):.4g}(n{len(v)})" for n, v in sorted(c.items())))
