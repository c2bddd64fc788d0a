"""Generates tests/golden/proofs_pyref.json: (circuit seed, toxic waste) -> vk JSON + proof JSON computed
by the independent big-int implementation oracle/pyref.py (keygen + prover), each proof checked by
pyref's pairing verifier (itself pinned on the reference's static triple, ref_static_triple.json).

The reference holds no (pk, witness) -> proof vector (its round-trip tests draw fresh toxic waste,
r1cs_gg_ppzksnark_zok.tcc:283-287), so these vectors pin the C oracle and, through it, the HIP path.
Run in the build container:  python tests/golden/make_golden_proofs.py
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import pyref
from ethsnarks_amd import r1cs as R

CASES = [  # (name, kind, nC, nIn, seed, small_values)
    ("random_m16", "random", 10, 2, 3, False),
    ("random_small_values_m64", "random", 40, 1, 40, True),
    ("chain_m32", "chain", 30, 1, R.SEED_DEFAULT, False),
    ("random_m128", "random", 100, 3, 100, False),
]
out = []
for name, kind, nC, nIn, seed, small in CASES:
    r, w = (R.synthetic_chain(nC, nIn, seed) if kind == "chain" else R.random_r1cs(nC, nIn, seed=seed, small_values=small))
    assert r.is_satisfied(w)
    rng = R.SplitMix64(seed ^ 0xA5A5A5A5)
    toxic = [rng.fr() for _ in range(5)]
    pk, vk = pyref.keygen(r.as_pyref(), *toxic)
    A, B, C = pyref.prove(pk, r.as_pyref(), w)
    proof = pyref.proof_to_json(A, B, C, w[1:1 + nIn])
    vkj = pyref.vk_to_json_dict(vk)
    assert pyref.verify(vkj, json.loads(proof)), name
    out.append(dict(name=name, kind=kind, nC=nC, nIn=nIn, seed=seed, small_values=small,
                    toxic=[hex(t) for t in toxic], m=r.domain_size, vk=vkj, proof_json=proof))
    print(name, "m =", r.domain_size, "ok")
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "proofs_pyref.json"), "w") as f:
    json.dump(out, f, indent=1)
