cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python3 tools/train_step_bench.py h 7b 2 > /tmp/tr.log 2>&1 < /dev/null
tail -1 /tmp/tr.log | cut -c1-300
timeout 120 python3 tools/probes/ktrace_summary.py /tmp/prof < /dev/null | head -16
