"""dev tool: small-circuit throughput with T host threads, each driving its own contexts (is the ~1 ms/proof floor a
per-thread submit limit or a GPU-side one?)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from ethsnarks_amd import prover as P, fields as F, gadgets as G
wl = sys.argv[1] if len(sys.argv) > 1 else "merkle29"
r, w, _ = G.merkle_membership_circuit(29) if wl == "merkle29" else G.mimc_preimage_circuit(11)
wm = F.fr_to_mont(w)
pk, vk = P.keygen(r, seed=3)
N = 300
def worker(ctxs, out, k):
    pending = []
    for i in range(N):
        if len(pending) == len(ctxs): pending.pop(0).collect()
        c = ctxs[i % len(ctxs)]; c.submit(wm); pending.append(c)
    while pending: pending.pop(0).collect()
for T in (1, 2, 3, 4):
    sets = [[P.ProverContext(pk, r) for _ in range(3)] for _ in range(T)]
    for cs in sets:
        for c in cs: c.submit(wm); c.collect()
    th = [threading.Thread(target=worker, args=(sets[k], None, k)) for k in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%s threads %d: %.0f proofs/s (%.3f ms per proof)" % (wl, T, T * N / dt, 1e3 * dt / (T * N)), flush=True)
    for cs in sets:
        for c in cs: c.close()
