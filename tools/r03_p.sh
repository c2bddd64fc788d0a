for q in 4 5 6 7 8; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python tools/dev_queue_map.py 20 0 2>&1 | grep logm; done
for q in 4 6; do for rep in 1 2; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 2^20 hwq $q', d['value'], d['ms_per_step'])"
done; done
for pr in hllhl hhhhh nnnnn; do echo "ZK_PRIOS=$pr"; ZK_PRIOS=$pr python tools/dev_sync_latency.py merkle29 14 16 18 20 2>&1 | grep sync; done
