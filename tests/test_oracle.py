"""CPU suite: pins the oracle chain (no GPU).

  reference static triple --validates--> pyref pairing verifier --accepts--> oracle proofs
  pyref big-int keygen/prover == C oracle keygen/prover, byte for byte (tests/golden/proofs_pyref.json)
"""
import json
import os
import numpy as np
import pytest
import pyref
from ethsnarks_amd import r1cs as R, fields as F
from helpers import GOLDEN, golden_cases, build_case, rand_scalars


def test_pyref_verifier_accepts_reference_static_triple():
    d = json.load(open(os.path.join(GOLDEN, "ref_static_triple.json")))
    assert pyref.verify(d["vk"], d["proof"])


def test_pyref_verifier_rejects_tampered_triple():
    d = json.load(open(os.path.join(GOLDEN, "ref_static_triple.json")))
    p = dict(d["proof"]); p["input"] = [p["input"][0], "0x8"]
    assert not pyref.verify(d["vk"], p)
    p = dict(d["proof"]); p["A"] = d["proof"]["C"]
    assert not pyref.verify(d["vk"], p)
    assert not pyref.verify(d["vk"], dict(d["proof"], input=d["proof"]["input"][:1]))   # strong IC


def test_constants():
    assert pow(pyref.ROOT_2_28, 1 << 28, pyref.R) == 1 and pow(pyref.ROOT_2_28, 1 << 27, pyref.R) == pyref.R - 1
    assert pyref.g1_on_curve(pyref.G1_GEN) and pyref.g2_on_curve(pyref.G2_GEN)
    assert pyref.g2_mul(pyref.G2_GEN, pyref.R) is None


def test_field_and_ntt_vs_pyref(oracle):
    rng = R.SplitMix64(1)
    a = [rng.fr() for _ in range(64)]; b = [rng.fr() for _ in range(64)]
    out = np.zeros((64, 4), dtype=np.uint64)
    oracle.lib().orc_fr_mul(oracle._p64(out), oracle._p64(F.fr_to_mont(a)), oracle._p64(F.fr_to_mont(b)), 64)
    assert F.fr_from_mont(out) == [x * y % F.FR for x, y in zip(a, b)]
    for logm in (0, 1, 4, 7):
        x = [rng.fr() for _ in range(1 << logm)]
        xm = F.fr_to_mont(x)
        assert F.fr_from_mont(oracle.ntt(xm, logm)) == pyref.ntt(x, pyref.omega(1 << logm))
        assert F.fr_from_mont(oracle.ntt(xm, logm, inverse=True)) == pyref.intt(x)
        assert F.fr_from_mont(oracle.ntt(xm, logm, coset=True)) == pyref.coset_ntt(x, 5)
        assert F.fr_from_mont(oracle.ntt(xm, logm, inverse=True, coset=True)) == pyref.coset_intt(x, 5)


def test_msm_bucket_vs_naive(oracle):
    for g2 in (False, True):
        n = 200
        sc = rand_scalars(n, 3, ones_every=5, zeros_every=7)
        sc[11] = F.FR - 1
        bases = oracle.batch_mul(F.fr_to_mont(rand_scalars(n, 4)), g2=g2)
        bases[20] = 0                      # infinity base
        bases[21] = bases[22]              # duplicate base
        s = F.fr_to_mont(sc)
        ref = oracle.msm(bases, s, g2=g2, naive=True)
        for c in (0, 3, 8):
            assert np.array_equal(oracle.msm(bases, s, g2=g2, c=c), ref)


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_oracle_matches_pyref_golden(oracle, case):
    r, w, toxic = build_case(case)
    assert r.domain_size == case["m"]
    pk, vk = oracle.keygen(r, toxic=toxic)
    assert json.loads(vk.to_json()) == case["vk"]
    proof_json, _ = oracle.prove(pk, r, F.fr_to_mont(w))
    assert proof_json == case["proof_json"]


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_trapdoor_closed_form_equals_the_provers(oracle, case):
    """The proof in closed form from the toxic waste (tcc:533-540 comments: A = (alpha + sum w_i A_i(t)) G1, ...): three scalar
    multiplications, no multi-exponentiation / transform / key / codec.  pyref's and the C oracle's closed forms equal each
    other and the golden proof both provers produce through the key."""
    r, w, toxic = build_case(case)
    A, B, C = pyref.proof_from_trapdoor(r.as_pyref(), w, *toxic)
    js = pyref.proof_to_json(A, B, C, w[1:1 + r.nIn])
    assert js == case["proof_json"]
    assert oracle.proof_from_trapdoor(r, F.fr_to_mont(w), toxic) == case["proof_json"]


def test_trapdoor_closed_form_on_seeded_keys_and_real_circuits(oracle):
    """seeded keys (what the GPU tests and bench.py use) and a circuit with general coefficients, unused variables and zero rows"""
    from ethsnarks_amd import gadgets as G
    for r, w, seed in [R.random_r1cs(200, 3, seed=12) + (21,), R.synthetic_chain((1 << 9) - 2, 1) + (R.SEED_DEFAULT,),
                       G.mimc_preimage_circuit(1)[:2] + (9,)]:
        wm = F.fr_to_mont(w)
        pk, _ = oracle.keygen(r, seed=seed)
        js, _ = oracle.prove(pk, r, wm)
        assert oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(seed)) == js
    # an unsatisfying witness has no proof through the key (degree check) and the closed form is not it either: nothing to compare,
    # but another toxic waste must give another proof
    assert oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(seed + 1)) != js


def test_oracle_proofs_verify_and_witness_map_matches(oracle):
    r, w = R.random_r1cs(24, 2, seed=77)
    wm = F.fr_to_mont(w)
    assert F.fr_from_mont(oracle.witness_map(r, wm)) == pyref.witness_map(r.as_pyref(), w)
    pk, vk = oracle.keygen(r, seed=5)
    js, _ = oracle.prove(pk, r, wm)
    assert pyref.verify(json.loads(vk.to_json()), json.loads(js))
    bad = json.loads(js); bad["input"][0] = "0x1"
    assert not pyref.verify(json.loads(vk.to_json()), bad)


def test_raw_roundtrip_and_layout(oracle, tmp_path):
    r, w = R.random_r1cs(12, 1, seed=9)
    pk, _ = oracle.keygen(r, seed=6)
    path = str(tmp_path / "pk.raw")
    pk.write_raw(path)
    raw = open(path, "rb").read()
    # first point = '0' + 64 raw bytes (Montgomery limbs of X, Y), no separators
    P = pk.parts()
    assert raw[0:1] == b"0" and raw[1:65] == P["alpha_g1"].tobytes()
    assert raw[65:66] == b"0" and raw[130:131] == b"0" and raw[131:259] == P["beta_g2"].tobytes()
    pk2 = oracle.read_raw(path)
    P2 = pk2.parts()
    for k in P:
        assert np.array_equal(np.asarray(P[k]), np.asarray(P2[k])), k
    # L query of this circuit contains points at infinity (unused variables): flag '1', X = 0, Y = R mod q
    assert any((P["L"][i] == 0).all() for i in range(pk.nL))
    with pytest.raises(IOError):
        oracle.read_raw(str(tmp_path / "missing.raw"))


def test_synthetic_chain_shape():
    r, w = R.synthetic_chain(30, 1)
    assert r.V == 32 and r.domain_size == 32 and r.is_satisfied(w)
    assert r.A.nnz == 15 * 2 + 14 + 31 and r.B.nnz == 29 + 31 and r.C.nnz == 30
    r2, _ = R.synthetic_chain((1 << 10) - 2, 1)
    assert r2.domain_size == 1 << 10


def test_r1cs_and_witness_json_dumps_roundtrip():
    """schema of src/export.cpp:157-221 (r1cs2json / witness2json)"""
    from ethsnarks_amd import gadgets as G
    r, w, _ = G.merkle_membership_circuit(1, leaf=5, address=1, path=[7])      # general coefficients, unique indices per row
    txt = R.r1cs_to_json(r)
    assert txt.startswith('{\n "nPubInputs": 1,\n "nOutputs": 0,\n "nVars": %d,\n "nConstraints": 737,\n "constraints": [\n  [{' % (r.V + 1))
    r2 = R.r1cs_from_json(txt)
    assert (r2.nC, r2.nIn, r2.V) == (r.nC, r.nIn, r.V)
    assert r2.A.to_rows() == r.A.to_rows() and r2.B.to_rows() == r.B.to_rows() and r2.C.to_rows() == r.C.to_rows()
    wt = R.witness_to_json(w)
    assert wt.startswith('[\n "1",\n "') and R.witness_from_json(wt) == w
    assert r2.is_satisfied(R.witness_from_json(wt))
