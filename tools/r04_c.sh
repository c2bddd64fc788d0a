#!/bin/bash
# round 4, call C: GPU parity suite (frugal tables, merged tail), the FP64 multiplier record with the device in RTZ mode, the synchronous
# proof after the per-call tail rule, and a rehearsal of bench.py's N > 1 legs (option-2 latency leg included) with 6 ranks on this one GPU over gloo
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?
tail -n 6 $O/gpu_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 ./tools/microbench > $O/microbench.txt 2> $O/microbench.err && head -n 9 $O/microbench.txt
timeout -k 10 600 ./tools/mulbench r04 > $O/mulbench.txt 2> $O/mulbench.err || { tail $O/mulbench.err; exit 1; }
grep -i "f52\|fp52\|dot2\|mul chain" $O/mulbench.txt
echo "== synchronous zk_prove" | tee $O/sync.txt
timeout -k 10 600 python tools/dev_sync_latency.py merkle29 16 18 20 2>&1 | tee -a $O/sync.txt
echo "== bench default" 
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d.get('launches_per_proof'), d['roofline']['traffic_source'], d.get('witness_modes',{}).get('resident',{}).get('value'))"
echo "== N = 6 rehearsal on one GPU (gloo, host-staged exchanges)"
ZK_BENCH_REHEARSE=1 timeout -k 10 1100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 6 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 6 --steps 3 --warmup 1 --extras-timeout 900 > $O/bench_n6_rehearsal.json 2> $O/bench_n6_rehearsal.err; rc=$?
tail -n 3 $O/bench_n6_rehearsal.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_n6_rehearsal.json").read().strip().splitlines()[-1])
    print("N6 value", d["value"], {k: d.get(k) for k in ("msm_sharded", "msm_sharded_latency", "msm_sharded_2p22", "extras_aborted")})
except Exception as e:
    print("no N6 line:", e)
PY
exit $rc
