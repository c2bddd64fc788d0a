R=$GRAFT_REPO_ROOT
for v in r02 head ntt512 default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  echo "== $v"; ZK_LIB_ALLOW_OLD_ABI=1 ZK_LIB=$L python tools/dev_sync_latency.py 16 17 2>&1 | grep sync
done
for v in r02 default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  echo "== $v serial per kernel 2^16"; ZK_LIB_ALLOW_OLD_ABI=1 ZK_LIB=$L python tools/dev_kernel_exclusive.py 16 2>&1 | grep -v amdgpu | head -22
done
