"""dev tool: per-kernel time of ONE proof with every launch on one stream (ZK_SERIAL=1), i.e. each kernel with the
machine to itself -- the work the pipelined schedule has to fit, as opposed to bench.py's kernel_sum (launches that
overlap each other).  Usage: python tools/dev_kernel_exclusive.py [logm] [workload chain|merkle29|mimc11]"""
import os, sys
os.environ["ZK_SERIAL"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
torch.cuda.init(); torch.zeros(1).cuda()          # torch's runtime first (see bench.py)
import numpy as np
from ethsnarks_amd import prover as P, r1cs as R, fields as F
P.load_library(os.environ.get("ZK_LIB") or None)          # ZK_LIB: A/B of two builds of the library
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = sys.argv[2] if len(sys.argv) > 2 else "chain"
if wl == "chain":
    r, w = R.synthetic_chain((1 << logm) - 2, 1)
else:
    from ethsnarks_amd import gadgets as G
    r, w, _ = G.merkle_membership_circuit(29) if wl == "merkle29" else G.mimc_preimage_circuit(11)
wm = F.fr_to_mont(w)
pk, vk = P.keygen(r, seed=R.SEED_DEFAULT)
ctx = P.ProverContext(pk, r)
for _ in range(3):
    ctx.prove_struct(wm)
P.profile_begin()
ctx.submit(wm); _, t = ctx.collect()
s_ms, n_l, per = P.profile_end()
print("one proof, serial: %.3f ms in %d launches; gpu_total %.3f ms" % (s_ms, n_l, t.get("gpu_total", float("nan"))))
for k, (c, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s calls %3d  %8.3f ms  %5.1f %%" % (k.strip("()"), c, v, 100 * v / s_ms))
print("phase timings (ms):", " ".join("%s=%.3f" % kv for kv in t.items()))
print("ctx info:", ctx.info())
