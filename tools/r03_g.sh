R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_g
mkdir -p $O
./tools/mulbench_nox2 nox2 > $O/mulbench.txt 2>&1
./tools/mulbench_x2 x2 >> $O/mulbench.txt 2>&1
cat $O/mulbench.txt
for v in g2only full g1w3; do
  ZK_LIB=$R/variants/$v/libzkhip.so python - > $O/parity_$v.txt 2>&1 <<PYEOF
import os, sys
sys.path[:0] = ["$R", "$R/tests", "$R/oracle"]
import numpy as np
import oracle_lib as O
from ethsnarks_amd import prover as P, r1cs as R, fields as F
from helpers import rand_scalars, tiled_bases
P.load_library(os.environ["ZK_LIB"])
ok = True
for g2 in (False, True):
    for n in (33, 1000, 20000, 1 << 16):
        sc = rand_scalars(n, 5, ones_every=9, zeros_every=11)
        bases = tiled_bases(O, n, g2=g2)
        s = F.fr_to_mont(sc)
        ok &= bool(np.array_equal(P.msm(bases, s, g2=g2), O.msm(bases, s, g2=g2)))
r, w = R.synthetic_chain((1 << 14) - 2, 1)
wm = F.fr_to_mont(w)
pk, _ = P.keygen(r, seed=5)
ctx = P.ProverContext(pk, r)
ok &= P.prove(ctx, wm) == O.prove(O.pk_from_parts(pk.parts()), r, wm)[0]
print("PARITY", "$v", ok)
PYEOF
  tail -1 $O/parity_$v.txt
done
for v in base g2only full g1w3; do
  ZK_LIB=$R/variants/$v/libzkhip.so python tools/dev_kernel_exclusive.py 20 > $O/excl_$v.txt 2>&1
  echo "== $v"; grep "accumulate\|one proof" $O/excl_$v.txt
done
for rep in 1 2; do for v in base g2only full g1w3; do
  ZK_LIB=$R/variants/$v/libzkhip.so python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
