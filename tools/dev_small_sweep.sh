# dev tool: sweep the MSM shape knobs on small circuits; prints one line per setting
# usage: LOGMS="12 15" SEGS="8" QUADS="0 1" QACCS="0 1" INFS="1 3" bash tools/dev_small_sweep.sh
for lm in ${LOGMS:-12 15 18}; do for inf in ${INFS:-1}; do for seg in ${SEGS:-8}; do for q in ${QUADS:-0 1}; do for qa in ${QACCS:-0 1}; do
ZK_SEG_MIN=$seg ZK_MSM_QUAD=$q ZK_MSM_QUAD_ACC=$qa timeout -k 10 200 python bench.py --logm $lm --inflight $inf --multi-exp-c ${C:-0} --steps 30 --warmup 5 --no-cpu-baseline 2>> ${GRAFT_REPO_ROOT:-.}/gpurun_out/dev_small_sweep.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms_last_step']; print('logm $lm inflight $inf seg $seg quad $q quad_acc $qa', 'ms', d['ms_per_step'], 'b_query', p['b_query'], 'acc_b', p['acc_b'], 'a', p['a_query'], 'h', p['h_query'], flush=True)" || exit 1
done; done; done; done; done
