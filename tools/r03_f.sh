R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03_f
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or config3 or chain_vs_oracle or random_circuits" > gpurun_out/r03_f/pytest.txt 2>&1
tail -3 gpurun_out/r03_f/pytest.txt
echo "== default (H early)" > gpurun_out/r03_f/sync.txt
python tools/dev_sync_latency.py 16 18 20 >> gpurun_out/r03_f/sync.txt 2>&1
echo "== ZK_H_EARLY=0" >> gpurun_out/r03_f/sync.txt
ZK_H_EARLY=0 python tools/dev_sync_latency.py 16 18 20 >> gpurun_out/r03_f/sync.txt 2>&1
echo "== default again" >> gpurun_out/r03_f/sync.txt
python tools/dev_sync_latency.py 18 20 >> gpurun_out/r03_f/sync.txt 2>&1
cat gpurun_out/r03_f/sync.txt
for i in 1 2; do
ZK_H_EARLY=0 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('H_EARLY=0', d['value'], d['ms_per_step'])"
python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default  ', d['value'], d['ms_per_step'])"
done
