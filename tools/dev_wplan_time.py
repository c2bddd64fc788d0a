"""dev tool: zk_wplan_solve time for the depth-29 Merkle circuit at several k (witnesses completed per call)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from ethsnarks_amd import prover as P, fields as F, gadgets as G
P.load_library(os.environ.get("ZK_LIB") or None)
r, w, _ = G.merkle_membership_circuit(29)
supplied = list(range(0, 1 + 1 + 29 + 29 + 1 + 29))
plan = P.WitnessPlan(r, supplied)
full = F.fr_to_mont(w)
for k in (1, 64, 256, 1024):
    start = np.zeros((k, r.V + 1, 4), dtype=np.uint64)
    start[:, supplied] = full[supplied]
    buf = P.DeviceBuffer(32 * (r.V + 1) * k)
    buf.upload(start)
    plan.solve(buf.ptr, k)
    buf.upload(start)
    t0 = time.perf_counter(); bad = plan.solve(buf.ptr, k); dt = time.perf_counter() - t0
    got = buf.download((k, r.V + 1, 4))
    ok = bad == 0 and all(np.array_equal(got[p], full) for p in (0, k - 1))
    print("k = %4d: %.1f ms (%.0f witnesses/s)  ok=%s" % (k, 1e3 * dt, k / dt, ok), flush=True)
    buf.free()
