R=$GRAFT_REPO_ROOT
for pad in "" 4 6 8 12; do
  echo "== pad '$pad'"; env ${pad:+ZK_STREAM_PAD=$pad} python tools/dev_sync_latency.py merkle29 16 20 2>&1 | grep sync
done
for rep in 1 2; do for pad in "" 2 4 6 8; do
  env ${pad:+ZK_STREAM_PAD=$pad} python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 2^20 pad $pad', d['value'], d['ms_per_step'])"
done; done
for pad in "" 4 8; do
  env ${pad:+ZK_STREAM_PAD=$pad} python bench.py --workload merkle29 --steps 300 --warmup 20 --witness resident --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench merkle29 pad $pad', d['value'], d['ms_per_step'])"
  env ${pad:+ZK_STREAM_PAD=$pad} python bench.py --logm 18 --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 2^18 pad $pad', d['value'], d['ms_per_step'])"
done
