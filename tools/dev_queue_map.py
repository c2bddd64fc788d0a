"""dev tool: how much the runtime's stream -> hardware-queue mapping matters.  One process: optional torch initialisation first (as
bench.py does), ZK_STREAM_PAD taken from the environment, then (a) synchronous proofs, (b) three contexts pipelined (host witness,
plain submit), at 2^logm.  usage: dev_queue_map.py [logm] [torch 0|1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 18
use_torch = len(sys.argv) > 2 and sys.argv[2] == "1"
if use_torch:
    import torch
    torch.cuda.set_device(0); torch.zeros(1).cuda(); torch.cuda.synchronize()
import numpy as np
from ethsnarks_amd import prover as P, r1cs as R, fields as F
P.load_library(os.environ.get("ZK_LIB") or None)
r, w = R.synthetic_chain((1 << logm) - 2, 1)
wm = F.fr_to_mont(w)
pk, _ = P.keygen(r, seed=3)
ctxs = [P.ProverContext(pk, r) for _ in range(3)]
for c in ctxs: c.prove_struct(wm)
ts = []
for _ in range(40):
    t = time.perf_counter(); ctxs[0].prove_struct(wm); ts.append(time.perf_counter() - t)
sync_ms = 1e3 * float(np.median(ts))
def run(n):
    pend = []
    for i in range(n):
        if len(pend) == 3: pend.pop(0).collect()
        c = ctxs[i % 3]; c.submit(wm); pend.append(c)
    while pend: pend.pop(0).collect()
run(6)
n = 60 if logm <= 18 else 30
t0 = time.perf_counter(); run(n); dt = time.perf_counter() - t0
print("logm %d torch %d pad %-4s: sync %.3f ms   pipelined %.3f ms/proof (%.1f proofs/s)" % (logm, use_torch, os.environ.get("ZK_STREAM_PAD", "-"), sync_ms, 1e3 * dt / n, n / dt), flush=True)
