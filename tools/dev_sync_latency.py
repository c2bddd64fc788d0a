"""dev tool: wall time of the synchronous zk_prove (what ethsnarks::prove maps to) for the two small circuits and two chain sizes"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from ethsnarks_amd import prover as P, r1cs as R, fields as F, gadgets as G
P.load_library(os.environ.get("ZK_LIB") or None)
for name in sys.argv[1:] or ["merkle29", "mimc11", "14", "16"]:
    if name == "merkle29": r, w, _ = G.merkle_membership_circuit(29)
    elif name == "mimc11": r, w, _ = G.mimc_preimage_circuit(11)
    else: r, w = R.synthetic_chain((1 << int(name)) - 2, 1)
    wm = F.fr_to_mont(w)
    pk, vk = P.keygen(r, seed=3)
    ctx = P.ProverContext(pk, r, latency=os.environ.get('ZK_NO_LATENCY_SCHED') != '1')       # what ethsnarks::prove's context is created with
    for _ in range(5): ctx.prove_struct(wm)
    ts = []
    for _ in range(60):
        t = time.perf_counter(); ctx.prove_struct(wm); ts.append(time.perf_counter() - t)
    print("%-9s sync prove: median %.3f ms  min %.3f ms" % (name, 1e3 * np.median(ts), 1e3 * np.min(ts)), flush=True)
    ctx.close(); pk.close()
