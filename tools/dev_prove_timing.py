"""dev tool: time zk_prove phases at a given size with an oracle-generated key (NOT the bench)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
from ethsnarks_amd import prover as P, r1cs as R, fields as F
import oracle_lib as O
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = time.time(); r, w = R.synthetic_chain((1 << logm) - 2, 1); wm = F.fr_to_mont(w); print("circuit %.1fs" % (time.time() - t), flush=True)
t = time.time(); pk_o, _ = O.keygen(r, seed=1); print("oracle keygen %.1fs (threads %d)" % (time.time() - t, O.lib().orc_num_threads()), flush=True)
t = time.time(); expect, ph = O.prove(pk_o, r, wm); print("oracle prove %.2fs phases %s" % (time.time() - t, ["%.3f" % x for x in ph]), flush=True)
pk = P.ProvingKey.from_parts(**pk_o.parts())
t = time.time(); ctx = P.ProverContext(pk, r, multi_exp_c=c); print("ctx create %.2fs" % (time.time() - t), flush=True)
for i in range(reps):
    t = time.time(); proof, tm = ctx.prove_struct(wm, timings=True); dt = time.time() - t
    print("prove %d: wall %.2f ms " % (i, dt * 1e3) + " ".join("%s=%.2f" % kv for kv in tm.items()), flush=True)
print("parity:", P.proof_to_json(proof, wm[1:2]) == expect)
