"""dev tool: host-side cost of zk_prove_submit / zk_prove_collect for a small circuit (is throughput launch-bound?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from ethsnarks_amd import prover as P, r1cs as R, fields as F, gadgets as G
wl = sys.argv[1] if len(sys.argv) > 1 else "merkle29"
r, w, _ = G.merkle_membership_circuit(29) if wl == "merkle29" else G.mimc_preimage_circuit(11)
wm = F.fr_to_mont(w)
pk, vk = P.keygen(r, seed=3)
ctxs = [P.ProverContext(pk, r) for _ in range(3)]
for c in ctxs: c.submit(wm); c.collect()
ts, tc = [], []
pending = []
t_all = time.perf_counter()
N = 300
for i in range(N):
    if len(pending) == 3:
        t = time.perf_counter(); pending.pop(0).collect(); tc.append(time.perf_counter() - t)
    c = ctxs[i % 3]
    t = time.perf_counter(); c.submit(wm); ts.append(time.perf_counter() - t)
    pending.append(c)
while pending:
    t = time.perf_counter(); pending.pop(0).collect(); tc.append(time.perf_counter() - t)
tot = time.perf_counter() - t_all
print("%s: %.3f ms/proof; submit host %.3f ms avg (min %.3f); collect host %.3f ms avg (min %.3f)" % (
    wl, 1e3 * tot / N, 1e3 * np.mean(ts), 1e3 * np.min(ts), 1e3 * np.mean(tc), 1e3 * np.min(tc)))
