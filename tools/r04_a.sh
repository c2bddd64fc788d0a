#!/bin/bash
# round 4, call A: GPU parity suite on the merged H + L tail, then a same-box A/B of the merged tail (bench + synchronous proof)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_a
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?
tail -n 5 $O/gpu_tests.txt
[ $rc -ne 0 ] && exit $rc
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], 'launches', d.get('launches_per_proof'), 'g2', d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
  python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2> $O/bench_merged_$rep.err | tee $O/bench_merged_$rep.json | line merged || exit 1
  ZK_NO_MERGE_HL=1 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2> $O/bench_separate_$rep.err | tee $O/bench_separate_$rep.json | line separate || exit 1
done
echo "== synchronous zk_prove, merged tail" | tee $O/sync.txt
python tools/dev_sync_latency.py merkle29 16 18 20 2>&1 | tee -a $O/sync.txt
echo "== synchronous zk_prove, separate tails (ZK_NO_MERGE_HL=1)" | tee -a $O/sync.txt
ZK_NO_MERGE_HL=1 python tools/dev_sync_latency.py merkle29 16 18 20 2>&1 | tee -a $O/sync.txt
