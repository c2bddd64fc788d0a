#!/bin/bash
# dev tool (GPU box): marginal cost of each phase INSIDE the full pipeline -- bench.py with the measurement build of the library
# (tools/build_variant.sh marginal -DZK_EXP_MARGINAL) dropping families of launches after the warm-up (the bench proves the same
# witness every step, so a proof that finds the previous proof's sorted digits / h / bucket sums in its buffers costs what it costs
# WITHOUT the dropped kernels).  usage: tools/dev_marginal.sh [steps]      (prints one line per experiment)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
STEPS=${1:-40}
cd $R
LOG=${R}/gpurun_out/dev_marginal.err; mkdir -p $(dirname $LOG); : > $LOG
# stderr is KEPT (round 3 sent it to /dev/null and lost the text of a device fault): one log for the whole sweep, and a run that prints no
# JSON line ends the sweep with its stderr on the screen -- no further GPU step after a failed one
run() {
  echo "== $1 (ZK_EXP_SKIP=$2)" >> $LOG
  ZK_EXP_SKIP="$2" ZK_LIB=$R/variants/marginal/libzkhip.so python bench.py --no-extras --no-cpu-baseline --witness resident --steps $STEPS --warmup 3 2>> $LOG \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-44s %8.3f ms/step  %7.2f proofs/s  launches/proof %.1f' % ('$1', d['ms_per_step'], d['value'], d['launches_per_proof']))" \
    || { echo "run '$1' failed; stderr of the sweep so far:"; tail -n 40 $LOG; exit 1; }
}
S="k_sort_"; P="k_spmv"; N="k_ntt_pass"; F="k_msm_bucket_finalize"; H="k_msm_heavy"; G="k_msm_rowcol_sum"; T="k_msm_weighted_sum"
A1="k_msm_accumulate<C, 1>"; A2="k_msm_accumulate<C, 2>"
run "everything" ""
run "no sorts" "$S"
run "no transforms" "$N"
run "no tails" "$F;$H;$G;$T"
run "accumulations only" "$S;$P;$N;$F;$H;$G;$T"
run "accumulations + sorts" "$P;$N;$F;$H;$G;$T"
run "accumulations + rows + transforms" "$S;$F;$H;$G;$T"
run "accumulations + finalize" "$S;$P;$N;$H;$G;$T"
run "accumulations + heavy" "$S;$P;$N;$F;$G;$T"
run "accumulations + rowcol_sum" "$S;$P;$N;$F;$H;$T"
run "accumulations + weighted_sum" "$S;$P;$N;$F;$H;$G"
run "accumulations + all tails" "$S;$P;$N"
run "G2 accumulation only" "$A2;$S;$P;$N;$F;$H;$G;$T"
run "G2 accumulation + sorts" "$A2;$P;$N;$F;$H;$G;$T"
run "G1 accumulations only" "$A1;$S;$P;$N;$F;$H;$G;$T"
run "G1 accumulations + sorts" "$A1;$P;$N;$F;$H;$G;$T"
run "everything" ""
