R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lane_layouts or config2" 2>&1 | tail -2
echo "== r02"; ZK_LIB_ALLOW_OLD_ABI=1 ZK_LIB=$R/variants/r02/libzkhip.so python tools/dev_sync_latency.py 16 17 18 19 20 2>&1 | grep sync
echo "== default"; python tools/dev_sync_latency.py 16 17 18 19 20 2>&1 | grep sync
echo "== pairs forced on"; ZK_ACC_PAIRS=1 python tools/dev_sync_latency.py 18 19 2>&1 | grep sync
echo "== pairs forced off"; ZK_ACC_PAIRS=0 python tools/dev_sync_latency.py 18 19 20 2>&1 | grep sync
for lm in 18 19; do for pr in 0 1; do
ZK_ACC_PAIRS=$pr python bench.py --logm $lm --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('logm $lm pairs $pr', d['value'], d['ms_per_step'])"
done; done
