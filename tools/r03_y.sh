for rep in 1 2; do for inf in 3 4; do
python bench.py --steps 30 --warmup 5 --inflight $inf --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $inf', d['value'], d['ms_per_step'], d['phases_ms_last_step'].get('compute_h'))"
done; done
python tools/dev_kernel_exclusive.py 20 2>&1 | grep "spmv\|one proof\|compute_h"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain_vs_oracle or long_rows or config3" 2>&1 | tail -1
