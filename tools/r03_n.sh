R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in r02 c1 c2 head default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  echo "== $v"; ZK_LIB_ALLOW_OLD_ABI=1 ZK_LIB=$L python tools/dev_sync_latency.py 17 18 2>&1 | grep sync
done; done
