#!/bin/bash
# round 4, call D: same-box sweeps of the schedule knobs after the merged tail, then a rehearsal of bench.py's N > 1 legs with 4 ranks on this one GPU
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_d
mkdir -p $O
cd $R
run() {  # name, env assignments..., then -- bench args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>> $O/sweep.err \
    | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %8.2f proofs/s  %6.3f ms/step  launches %.1f' % ('$name', d['value'], d['ms_per_step'], d['launches_per_proof']))" \
    | tee -a $O/sweep.txt || { echo "run $name failed"; tail -n 20 $O/sweep.err; exit 1; }
}
: > $O/sweep.txt; : > $O/sweep.err
run default X=1 --
run hl_tail_on_l ZK_HL_TAIL=l --
run hl_tail_on_a ZK_HL_TAIL=a --
run b_tail_4_lanes ZK_B_TAIL_LANES=4 --
run default X=1 --
run group_16 ZK_MSM_GROUP=16 --
run group_4 ZK_MSM_GROUP=4 --
run seg_max_128 ZK_SEG_MAX=128 --
run seg_max_48 ZK_SEG_MAX=48 --
run default X=1 --
run inflight_4 X=1 -- --inflight 4
run inflight_2 X=1 -- --inflight 2
run prios_hlhhl ZK_PRIOS=hlhhl --
run prios_hllll ZK_PRIOS=hllll --
run default X=1 --
echo "== N = 4 rehearsal on one GPU (gloo, host-staged exchanges)"
ZK_BENCH_REHEARSE=1 timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 3 --warmup 1 --extras-timeout 800 > $O/bench_n4_rehearsal.json 2> $O/bench_n4_rehearsal.err; rc=$?
tail -n 3 $O/bench_n4_rehearsal.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_n4_rehearsal.json").read().strip().splitlines()[-1])
    print("N4 value", d["value"], {k: d.get(k) for k in ("msm_sharded", "msm_sharded_latency", "msm_sharded_2p22", "extras_aborted")})
except Exception as e:
    print("no N4 line:", e)
PY
exit $rc
