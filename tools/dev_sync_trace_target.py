"""dev tool: 30 synchronous proofs of the Merkle-29 circuit, to be run under rocprofv3 --kernel-trace (timeline of one small proof)"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
from ethsnarks_amd import prover as P, fields as F, gadgets as G
r, w, _ = G.merkle_membership_circuit(29)
wm = F.fr_to_mont(w)
pk, vk = P.keygen(r, seed=3)
ctx = P.ProverContext(pk, r)
for _ in range(30): ctx.prove_struct(wm)
