"""dev tool: per-kernel time of one batched launch sequence (k witnesses of a small circuit), kernels overlapping as in production
usage: dev_batch_profile.py [merkle29|mimc11] [k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from ethsnarks_amd import prover as P, fields as F, gadgets as G
P.load_library(os.environ.get("ZK_LIB") or None)
wl = sys.argv[1] if len(sys.argv) > 1 else "merkle29"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
r, w, _ = G.merkle_membership_circuit(29) if wl == "merkle29" else G.mimc_preimage_circuit(11)
wm = F.fr_to_mont(w)
pk, vk = P.keygen(r, seed=3)
ctx = P.ProverContext(pk, r, max_batch=k)
ws = np.ascontiguousarray(np.tile(wm.reshape(1, -1, 4), (k, 1, 1)))
for _ in range(3):
    ctx.submit_batch(ws, k=k); ctx.collect_batch(k)
for serial in (False,):
    P.profile_begin()
    ctx.submit_batch(ws, k=k); _, t = ctx.collect_batch(k)
    s_ms, n_l, per = P.profile_end()
    print("%s k=%d: kernel sum %.3f ms in %d launches; gpu_total %.3f ms" % (wl, k, s_ms, n_l, t.get("gpu_total", float("nan"))))
    for kk, (c, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-44s calls %3d  %8.3f ms  %5.1f %%" % (kk.strip("()"), c, v, 100 * v / s_ms))
    print("phase timings (ms):", " ".join("%s=%.3f" % kv for kv in t.items()))
print("ctx info:", ctx.info())
