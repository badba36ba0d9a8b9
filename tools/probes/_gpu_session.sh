cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python bench.py --mode decode --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
