"""dev tool: N synchronous proofs (zk_prove: host witness in, proof out) of one circuit, to be run under
rocprofv3 --kernel-trace --memory-copy-trace; tools/sync_timeline.py prints the timeline of the last one.
usage: dev_sync_trace_target.py [merkle29 | mimc11 | <logm>] [n]"""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
from ethsnarks_amd import prover as P, fields as F, gadgets as G, r1cs as R
name = sys.argv[1] if len(sys.argv) > 1 else "merkle29"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if name == "merkle29": r, w, _ = G.merkle_membership_circuit(29)
elif name == "mimc11": r, w, _ = G.mimc_preimage_circuit(11)
else: r, w = R.synthetic_chain((1 << int(name)) - 2, 1)
wm = F.fr_to_mont(w)
pk, vk = P.keygen(r, seed=3)
ctx = P.ProverContext(pk, r, latency=os.environ.get('ZK_NO_LATENCY_SCHED') != '1')       # what ethsnarks::prove's context is created with
ts = []
for _ in range(n):
    t = time.perf_counter(); ctx.prove_struct(wm); ts.append(time.perf_counter() - t)
print("%s: sync prove median %.3f ms, last %.3f ms" % (name, 1e3 * float(np.median(ts[3:])), 1e3 * ts[-1]))
