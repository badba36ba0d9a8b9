cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in libullsam_hip.so libullsam_hip_prio1.so libullsam_hip_prio3.so; do
echo "== $lib"; ULLSAM_HIP_LIB=$GRAFT_REPO_ROOT/ullsam_amd/lib/$lib timeout 300 python3 tools/attn_bench.py 0 2>&1 < /dev/null | grep variant
done; done
