# dev tool: sweep the MSM shape knobs on small circuits (latency mode); prints one line per setting
for lm in ${LOGMS:-12 15 18}; do for c in ${CS:-0}; do for seg in ${SEGS:-32 16 8 4}; do for g in ${GROUPS_:-8 4 2 1}; do
ZK_SEG_MIN=$seg ZK_MSM_GROUP=$g timeout -k 10 200 python bench.py --logm $lm --inflight 1 --multi-exp-c $c --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms_last_step']; print('logm $lm c $c seg $seg group $g', 'ms', d['ms_per_step'], 'b_query', p['b_query'], 'acc_b', p['acc_b'], 'a', p['a_query'], 'h', p['h_query'], flush=True)" || exit 1
done; done; done; done
