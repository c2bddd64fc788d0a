"""GPU parity tests: libzkhip.so (HIP, gfx950) through its C ABI vs the CPU oracle on the same
seeded inputs; bit-exact (integer arithmetic, no tolerance).  Full-size cases use size-independent
properties (round trips, permutation / split invariance)."""
import json
import numpy as np
import pytest
from ethsnarks_amd import r1cs as R, fields as F
from helpers import golden_cases, build_case, rand_scalars, tiled_bases

pytestmark = pytest.mark.gpu


def test_native_library_is_loaded(hip):
    assert hip._lib_path_loaded.endswith("ethsnarks_amd/libzkhip.so")
    assert hip.device_count() >= 1


def test_field_mul(hip, oracle):
    n = 5000
    a = F.fr_to_mont(rand_scalars(n, 1)); b = F.fr_to_mont(rand_scalars(n, 2))
    a[0] = 0; b[1] = 0; a[2] = F.ints_to_limbs([F.FR - 1])[0]; b[2] = a[2]
    o = np.zeros_like(a)
    oracle.lib().orc_fr_mul(oracle._p64(o), oracle._p64(a), oracle._p64(b), n)
    assert np.array_equal(hip.field_mul(a, b, "fr"), o)
    a[2] = F.ints_to_limbs([F.FQ - 1])[0]; b[2] = a[2]
    oracle.lib().orc_fq_mul(oracle._p64(o), oracle._p64(a), oracle._p64(b), n)
    assert np.array_equal(hip.field_mul(a, b, "fq"), o)


@pytest.mark.parametrize("logm", [0, 1, 3, 4, 10, 11, 12, 13, 15, 18, 20, 21])   # stage counts per pass: odd and even, one and several pairs, two and three passes
def test_ntt_vs_oracle(hip, oracle, logm):
    rng = np.random.default_rng(logm)
    x = rng.integers(0, 1 << 62, size=(1 << logm, 4), dtype=np.uint64)
    x[:, 3] &= (1 << 59) - 1                        # < 2^251 < r: valid reduced representatives
    for inv in (False, True):
        for coset in (False, True):
            assert np.array_equal(hip.ntt(x, logm, inv, coset), oracle.ntt(x, logm, inv, coset)), (logm, inv, coset)


def test_ntt_roundtrip_full_size(hip):
    logm = 22
    rng = np.random.default_rng(7)
    x = rng.integers(0, 1 << 62, size=(1 << logm, 4), dtype=np.uint64)
    x[:, 3] &= (1 << 59) - 1
    y = hip.ntt(x, logm, False, True)
    assert not np.array_equal(x, y)
    assert np.array_equal(hip.ntt(y, logm, True, True), x)


@pytest.mark.parametrize("g2", [False, True], ids=["G1", "G2"])
@pytest.mark.parametrize("n,c", [(0, 0), (1, 0), (33, 0), (1000, 0), (1000, 11), (20000, 0), (1 << 16, 0)])
def test_msm_vs_oracle(hip, oracle, g2, n, c):
    sc = rand_scalars(n, n + 1, ones_every=5, zeros_every=7)
    if n > 20:
        sc[11] = F.FR - 1; sc[12] = 2; sc[13] = 1 << 253
    s = F.fr_to_mont(sc) if n else np.zeros((0, 4), dtype=np.uint64)
    bases = tiled_bases(oracle, n, g2=g2, distinct=4096)
    if n > 30:
        bases[20] = 0; bases[21] = bases[22]
    assert np.array_equal(hip.msm(bases, s, g2=g2, c=c), oracle.msm(bases, s, g2=g2))


def test_msm_heavy_buckets(hip, oracle):
    n = 40000
    s = F.fr_to_mont([1] * 30000 + [77] * 5000 + rand_scalars(5000, 5))
    bases = tiled_bases(oracle, n, distinct=4096)
    assert np.array_equal(hip.msm(bases, s), oracle.msm(bases, s))
    assert np.array_equal(hip.msm(bases, s, c=6), oracle.msm(bases, s))


def test_msm_exceptional_cases(hip, oracle):
    pts = oracle.batch_mul(F.fr_to_mont([5, 5, 9]))
    neg = pts[0].copy()
    y = F.fq_from_mont(neg[4:8].reshape(1, 4))[0]
    neg[4:8] = F.fq_to_mont([(-y) % F.FQ])[0]
    bases = np.stack([pts[0], neg, pts[2], pts[2]])
    s = F.fr_to_mont([3, 3, 4, 4])
    assert np.array_equal(hip.msm(bases, s, c=3), oracle.msm(bases, s, naive=True))
    assert (hip.msm(bases, F.fr_to_mont([3, 3, 0, 0]), c=3) == 0).all()


@pytest.mark.parametrize("quad,quad_acc,pairs", [("1", "1", "0"), ("1", "0", "0"), ("0", "0", "0"), ("0", "0", "1")], ids=["quad", "quad-tails", "one-lane", "one-lane-pairs"])
@pytest.mark.parametrize("g2", [False, True], ids=["G1", "G2"])
def test_msm_lane_layouts_and_segment_level_exceptions(hip, oracle, monkeypatch, g2, quad, quad_acc, pairs):
    """Both lane layouts of the accumulation / reduction kernels (four cooperating lanes per logical thread for small
    MSMs, one for large ones) on inputs whose SEGMENT sums collide: 16 copies of P under one scalar (8-entry segments ->
    8P + 8P, the doubling branch of the XYZZ addition), then 8 x P and 8 x -P (8P + -8P = infinity), plus ordinary data."""
    monkeypatch.setenv("ZK_MSM_QUAD", quad); monkeypatch.setenv("ZK_MSM_QUAD_ACC", quad_acc)
    monkeypatch.setenv("ZK_ACC_PAIRS", pairs)              # the mixed addition with dual-issue product pairs (what machine-filling G1 sizes use)
    pts = oracle.batch_mul(F.fr_to_mont([5, 9]), g2=g2)
    neg = pts[1].copy()
    k = 8 if g2 else 4                                     # limbs (u64) of one coordinate
    ycoords = F.fq_from_mont(neg[k:2 * k].reshape(-1, 4))
    neg[k:2 * k] = F.fq_to_mont([(-v) % F.FQ for v in ycoords]).reshape(-1)
    rest = tiled_bases(oracle, 600, g2=g2, distinct=256)
    bases = np.concatenate([np.tile(pts[0], (16, 1)), np.tile(pts[1], (8, 1)), np.tile(neg, (8, 1)), rest])
    sc = [12345] * 16 + [777] * 16 + rand_scalars(600, 17, ones_every=6, zeros_every=11)
    s = F.fr_to_mont(sc)
    for c in (0, 5, 11):
        assert np.array_equal(hip.msm(bases, s, g2=g2, c=c), oracle.msm(bases, s, g2=g2))


def test_msm_full_size_properties(hip, oracle):
    """2^20 points: result is invariant under a permutation of the (base, scalar) pairs and equals the
    oracle on the same data (the oracle takes a few seconds at this size with 16 threads)."""
    n = 1 << 20
    rng = np.random.default_rng(3)
    s = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); s[:, 3] &= (1 << 59) - 1
    bases = tiled_bases(oracle, n, distinct=8192)
    r1 = hip.msm(bases, s)
    perm = rng.permutation(n)
    assert np.array_equal(hip.msm(bases[perm], s[perm]), r1)
    assert np.array_equal(r1, oracle.msm(bases, s))


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_prove_golden_vectors(hip, oracle, case):
    r, w, toxic = build_case(case)
    pk_o, _ = oracle.keygen(r, toxic=toxic)
    pk = hip.ProvingKey.from_parts(**pk_o.parts())
    ctx = hip.ProverContext(pk, r)
    assert hip.prove(ctx, F.fr_to_mont(w)) == case["proof_json"]
    assert hip.prove(ctx, F.ints_to_limbs(w), canonical=True) == case["proof_json"]
    ctx.close()


@pytest.mark.parametrize("logm", [9, 12, 14])
def test_prove_chain_vs_oracle(hip, oracle, logm, tmp_path):
    r, w = R.synthetic_chain((1 << logm) - 2, 1)
    wm = F.fr_to_mont(w)
    pk_o, vk_o = oracle.keygen(r, seed=logm)
    expect, _ = oracle.prove(pk_o, r, wm)
    path = str(tmp_path / "pk.raw")
    pk_o.write_raw(path)
    pk = hip.load_proving_key(path)                     # the .raw reader is on the measured path's boundary
    ctx = hip.ProverContext(pk, r)
    assert np.array_equal(ctx.witness_map(wm), oracle.witness_map(r, wm))
    got = hip.prove(ctx, wm)
    assert got == expect
    assert hip.prove(ctx, wm) == expect                 # context reuse (ProverContext scratch, hpp:286-289)
    parts = [hip.ProverContext(pk, r, shard_rank=k, shard_count=4).prove_partial(wm) for k in range(4)]
    assert hip.proof_to_json(ctx.prove_combine(np.stack(parts)), wm[1:2]) == expect
    if logm == 9:
        import pyref
        assert pyref.verify(json.loads(vk_o.to_json()), json.loads(got))


def test_prove_random_circuits_vs_oracle(hip, oracle):
    for nC, nIn, small in ((50, 2, False), (300, 1, True), (3000, 5, False)):
        r, w = R.random_r1cs(nC, nIn, seed=nC, small_values=small)
        wm = F.fr_to_mont(w)
        pk_o, _ = oracle.keygen(r, seed=nC)
        pk = hip.ProvingKey.from_parts(**pk_o.parts())
        ctx = hip.ProverContext(pk, r)
        expect, _ = oracle.prove(pk_o, r, wm)
        assert hip.prove(ctx, wm) == expect
        one = hip.ProverContext(pk, r, one_stream=True)     # zk_config.schedule = ZK_SCHED_ONE_STREAM: same proof
        assert hip.prove(one, wm) == expect and hip.prove(one, wm) == expect


@pytest.mark.parametrize("nC,small,batch", [(20000, True, 1), (50000, False, 1), (9000, True, 3)])
def test_prove_mid_size_structured_witnesses_vs_oracle(hip, oracle, nC, small, batch):
    """mid-size random circuits (domains 2^14 ... 2^16: other window sizes and bucket matrices than the chain sizes), with 0 / 1 / 2 / 3-valued
    witnesses (one bucket holds most entries: the heavy-bucket path inside the merged H + L tail and the row / column reduction), pipelined
    (two tails merged) and synchronous (small proofs keep the L tail), and as a batch of different witnesses of one circuit"""
    r, w = R.random_r1cs(nC, 2, seed=nC, small_values=small, max_terms=3)
    wm = F.fr_to_mont(w)
    pk_o, _ = oracle.keygen(r, seed=nC + 1)
    pk = hip.ProvingKey.from_parts(**pk_o.parts())
    expect, _ = oracle.prove(pk_o, r, wm)
    ctx = hip.ProverContext(pk, r, max_batch=batch)
    assert hip.prove(ctx, wm) == expect                              # synchronous entry point
    ctx.submit(wm)                                                   # queued entry point (merged tail)
    part, _ = ctx.collect()
    assert hip.proof_to_json(ctx.prove_combine(part), wm[1:3]) == expect
    if batch > 1:
        ws = [wm] + [F.fr_to_mont(R.random_r1cs(nC, 2, seed=nC, small_values=small, max_terms=3, witness_seed=100 + i)[1]) for i in range(batch - 1)]
        assert hip.prove_batch(ctx, np.stack(ws)) == [oracle.prove(pk_o, r, x)[0] for x in ws]
    ctx.close(); pk.close()


def test_prove_shape_sweep_vs_oracle(hip, oracle):
    """domain-size boundaries (nC + nIn + 1 on either side of a power of two), no public inputs, one constraint, dense and
    sparse rows, unreferenced variables, 0/1-heavy witnesses: every proof byte-equal to the oracle's"""
    shapes = [(1, 0), (1, 1), (2, 0), (3, 2), (5, 1), (6, 1), (7, 0), (13, 2), (29, 2), (30, 1), (31, 0), (61, 2), (62, 1), (63, 0),
              (126, 1), (127, 0), (250, 5), (254, 1), (255, 0), (509, 2), (1000, 3), (1021, 2), (1022, 1), (1023, 0), (2046, 1), (2047, 0)]
    for i, (nC, nIn) in enumerate(shapes):
        r, w = R.random_r1cs(nC, nIn, n_extra_vars=i % 4, max_terms=1 + i % 8, seed=1000 + i, small_values=(i % 3 == 1))
        wm = F.fr_to_mont(w)
        pk_o, _ = oracle.keygen(r, seed=77 + i)
        expect, _ = oracle.prove(pk_o, r, wm)
        pk = hip.ProvingKey.from_parts(**pk_o.parts())
        ctx = hip.ProverContext(pk, r, one_stream=bool(i & 1))
        assert r.domain_size == hip.get_domain_size(r)
        assert hip.prove(ctx, wm) == expect, (nC, nIn)
        ctx.close(); pk.close()


def test_errors(hip, oracle, tmp_path):
    r, w = R.random_r1cs(12, 1, seed=2)
    pk_o, _ = oracle.keygen(r, seed=3)
    pk = hip.ProvingKey.from_parts(**pk_o.parts())
    ctx = hip.ProverContext(pk, r)
    bad = list(w); bad[-4] = (bad[-4] + 1) % F.FR
    with pytest.raises(hip.ZkError) as e:
        hip.prove(ctx, F.fr_to_mont(bad))
    assert e.value.code == 7
    with pytest.raises(hip.ZkError) as e:
        hip.load_proving_key(str(tmp_path / "nope.raw"))
    assert e.value.code == 2


def test_config1_mimc_hash_preimage_11_words(hip, oracle):
    """BASELINE config 1's workload on the HIP path: MiMC-e7 Miyaguchi-Preneel hash preimage over 11 message words
    (the in-tree substitute of the SHA256 hashpreimage gadget), 4 016 constraints, m = 2^12.  GPU keygen ==
    oracle keygen, proof byte-identical to the oracle, accepted by zk_verify and by the pinned big-int verifier."""
    import pyref
    from ethsnarks_amd import gadgets as G
    r, w, _ = G.mimc_preimage_circuit(11)
    assert r.domain_size == 1 << 12
    wm = F.fr_to_mont(w)
    pk, vk = hip.keygen(r, seed=11)
    pk_o, vk_o = oracle.keygen(r, seed=11)
    assert vk.to_json() == vk_o.to_json()
    ctx = hip.ProverContext(pk, r)
    got = hip.prove(ctx, wm)
    expect, _ = oracle.prove(pk_o, r, wm)
    assert got == expect
    assert got == oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(11))     # closed form: no MSM / NTT / key / codec involved
    assert hip.stub_verify(vk.to_json(), got)
    assert pyref.verify(json.loads(vk.to_json()), json.loads(got))
    bad = json.loads(got); bad["A"], bad["C"] = bad["C"], bad["A"]
    assert not hip.stub_verify(vk.to_json(), json.dumps(bad))
    ctx.close()


def test_json_writers_reproduce_the_reference_static_triple(hip):
    """The JSON WRITERS on text the reference holds (test/test_verify.py:10-12): the static proof and key are parsed into
    zk_proof / zk_vk (zk_proof_from_json / zk_vk_from_json = proof_from_json / vk_from_json, src/import.cpp:161-223) and
    written back by zk_proof_to_json / zk_vk_to_json (src/export.cpp:20-33,56-145): every "0x..." token (63-digit values
    included: no zero padding, lowercase) and the [c1, c0] nesting must come out as the fixture has them."""
    import os, re
    from helpers import GOLDEN
    d = json.load(open(os.path.join(GOLDEN, "ref_static_triple.json")))
    proof, inputs = hip.proof_from_json(json.dumps(d["proof"]))
    text = hip.proof_to_json(proof, inputs, canonical=True)
    assert json.loads(text) == d["proof"]
    flat = lambda x: [x] if isinstance(x, str) else [t for y in x for t in flat(y)]
    assert re.findall(r'"(0x[0-9a-f]+)"', text) == flat([d["proof"][k] for k in ("A", "B", "C", "input")])
    assert text.startswith('{\n "A" :["0x') and '],\n "B"  :[["0x' in text and '"],\n ["0x' in text and '],\n "input" :["0x' in text
    vk = hip.vk_from_json(json.dumps(d["vk"]))
    vtext = vk.to_json()
    assert json.loads(vtext) == d["vk"]
    assert re.findall(r'"(0x[0-9a-f]+)"', vtext) == flat([d["vk"][k] for k in ("alpha", "beta", "gamma", "delta", "gammaABC")])
    assert any(len(t) < 66 for t in flat(d["vk"])) and any(len(t) < 66 for t in flat(d["proof"]))      # the fixture does hold short values
    assert hip.stub_verify(vtext, text)


def test_config4_merkle_membership_depth29(hip, oracle):
    """BASELINE config 4: merkle_path_authenticator<MiMC_e7_hash_gadget>, depth 29, 21 345 constraints, m = 2^15;
    key from the GPU generator, proof byte-identical to the oracle and accepted by the pinned verifier"""
    import pyref
    from ethsnarks_amd import gadgets as G
    r, w, root = G.merkle_membership_circuit(29)
    assert root == 14972246236048249827985830600768475898195156734731557762844426864943654467818
    wm = F.fr_to_mont(w)
    pk, vk = hip.keygen(r, seed=29)
    pk_o, vk_o = oracle.keygen(r, seed=29)
    assert vk.to_json() == vk_o.to_json()                                   # GPU keygen == oracle keygen
    P1, P2 = pk.parts(), pk_o.parts()
    assert all(np.array_equal(np.asarray(P1[k]), np.asarray(P2[k])) for k in P1)
    ctx = hip.ProverContext(pk, r)
    got = hip.prove(ctx, wm)
    expect, _ = oracle.prove(pk_o, r, wm)
    assert got == expect
    assert got == oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(29))     # closed form from the toxic waste
    proof = json.loads(got)
    assert int(proof["input"][0], 16) == root
    assert pyref.verify(json.loads(vk.to_json()), proof)
    assert hip.stub_verify(vk.to_json(), got)                              # the product's own verifier (zk_verify)
    # a different leaf position: address bits / selector products are 0/1-valued witness entries
    r2, w2, _ = G.merkle_membership_circuit(29, leaf=12345, address=0x15555555, path=[G.merkle_unique(d, 7) for d in range(29)])
    expect2, _ = oracle.prove(pk_o, r2, F.fr_to_mont(w2))                   # same constraint system, same key
    assert hip.prove(ctx, F.fr_to_mont(w2)) == expect2 == oracle.proof_from_trapdoor(r2, F.fr_to_mont(w2), oracle.toxic_from_seed(29))


@pytest.mark.parametrize("k", [1, 4, 16])
def test_prove_batch_merkle29(hip, oracle, k):
    """k membership proofs of the depth-29 MiMC Merkle circuit (different leaves, addresses and paths: many
    0/1-valued witness entries) through ONE launch sequence; proof p is byte-identical to the oracle's proof of witness p"""
    from ethsnarks_amd import gadgets as G
    r, w0, _ = G.merkle_membership_circuit(29)
    pk, vk = hip.keygen(r, seed=29)
    pk_o = oracle.pk_from_parts(pk.parts())
    ws = [F.fr_to_mont(w0)]
    for p in range(1, k):
        _, w, _ = G.merkle_membership_circuit(29, leaf=1000 + p, address=(0x15555555 * (p + 1)) & ((1 << 29) - 1), path=[G.merkle_unique(d, 7 + p) for d in range(29)])
        ws.append(F.fr_to_mont(w))
    expect = [oracle.prove(pk_o, r, w)[0] for w in ws]
    ctx = hip.ProverContext(pk, r, max_batch=16)
    got = hip.prove_batch(ctx, np.stack(ws))
    assert got == expect
    assert all(hip.stub_verify(vk.to_json(), g) for g in got[:2])
    assert hip.prove(ctx, ws[-1]) == expect[-1]                    # single proofs on the same context
    ctx.close()


def test_witness_plan_merkle29_on_device_then_batch_prove(hip, oracle):
    """SURVEY 8(f)-4, the whole pipeline resident: the depth-29 Merkle circuit's constraint system compiled as a forward-
    substitution program (zk_wplan) completes k witnesses on the GPU from the supplied variables only (root, address bits, path,
    leaf, IVs), the buffer goes to zk_prove_batch_submit_resident, and every proof equals the oracle's proof of the witness the
    host front end computes for the same inputs."""
    import time
    from ethsnarks_amd import gadgets as G
    k = 4
    cases = [G.merkle_membership_circuit(29)] + [
        G.merkle_membership_circuit(29, leaf=2000 + p, address=(0x0f0f0f0f * (p + 3)) & ((1 << 29) - 1), path=[G.merkle_unique(d, 11 + p) for d in range(29)])
        for p in range(1, k)]
    r = cases[0][0]
    supplied = list(range(0, 1 + 1 + 29 + 29 + 1 + 29))             # ONE, expected_root, address_bits, path, leaf, IVs (allocation order)
    plan = hip.WitnessPlan(r, supplied)
    start = np.zeros((k, r.V + 1, 4), dtype=np.uint64)
    for p in range(k):
        start[p, supplied] = F.fr_to_mont([cases[p][1][i] for i in supplied])
    buf = hip.DeviceBuffer(32 * (r.V + 1) * k)
    buf.upload(start)
    t0 = time.perf_counter()
    assert plan.solve(buf.ptr, k) == 0
    print("zk_wplan_solve: %d Merkle-29 witnesses in %.1f ms" % (k, 1e3 * (time.perf_counter() - t0)))
    got = buf.download((k, r.V + 1, 4))
    for p in range(k):
        assert np.array_equal(got[p], F.fr_to_mont(cases[p][1]))
    pk, _ = hip.keygen(r, seed=29)
    pk_o = oracle.pk_from_parts(pk.parts())
    ctx = hip.ProverContext(pk, r, max_batch=k)
    ctx.submit_batch(None, device_ptr=buf.ptr, k=k)
    parts, _ = ctx.collect_batch(k)
    for p in range(k):
        assert hip.proof_to_json(ctx.prove_combine(parts[p]), got[p][1:2]) == oracle.prove(pk_o, r, got[p])[0]
    ctx.close(); plan.close(); buf.free()


def test_witness_plan_with_bit_decomposition_hints(hip, oracle):
    """Gadgets with non-deterministic advice (the reference's field2bits family, src/gadgets/field2bits_strict.cpp): the constraints only
    CHECK the bits, so the witness program needs a hint -- ZK_WHINT_BITS: w[first + i] = bit i of w[src].  253-bit decomposition (a
    253-term row: many tape records with carried sums), k witnesses completed from x alone, proofs equal to the oracle's."""
    from ethsnarks_amd import gadgets as G
    k = 4
    cases = [G.field2bits_circuit(253, seed=90 + p) for p in range(k)]
    r, (xv, first, nb) = cases[0][0], cases[0][2]
    supplied = [0, xv, first + nb + 1]                               # ONE, x, iv (allocation order: x, bits, y, iv, ...)
    with pytest.raises(hip.ZkError):
        hip.WitnessPlan(r, supplied)                                 # not in solved order without the hint
    plan = hip.WitnessPlan(r, supplied, bit_hints=[(xv, first, nb)])
    start = np.zeros((k, r.V + 1, 4), dtype=np.uint64)
    for p in range(k):
        start[p, supplied] = F.fr_to_mont([cases[p][1][i] for i in supplied])
    buf = hip.DeviceBuffer(32 * (r.V + 1) * k)
    buf.upload(start)
    assert plan.solve(buf.ptr, k) == 0
    got = buf.download((k, r.V + 1, 4))
    for p in range(k):
        assert np.array_equal(got[p], F.fr_to_mont(cases[p][1]))
    pk, _ = hip.keygen(r, seed=31)
    pk_o = oracle.pk_from_parts(pk.parts())
    ctx = hip.ProverContext(pk, r, max_batch=k)
    ctx.submit_batch(None, device_ptr=buf.ptr, k=k)
    parts, _ = ctx.collect_batch(k)
    for p in range(k):
        assert hip.proof_to_json(ctx.prove_combine(parts[p]), got[p][1:2]) == oracle.prove(pk_o, r, got[p])[0]
    ctx.close(); plan.close(); buf.free()


def test_witness_plan_with_inverse_and_nonzero_hints(hip, oracle):
    """The reference's IsNonZero gadget (src/gadgets/isnonzero.cpp:34-60): M = 1 / X (0 for 0) and Y = [X != 0] are advice its constraints only
    check (the first reads Y before anything defines it).  ZK_WHINT_INV / ZK_WHINT_NONZERO complete the witnesses on the GPU from the X alone;
    proofs of the completed witnesses equal the oracle's."""
    from ethsnarks_amd import gadgets as G
    vals = [(0, 5, 0, 7, 1), (3, 0, 0, 0, 9), (0, 0, 0, 0, 0), (1, 2, 3, 4, 5)]
    cases = [G.isnonzero_circuit(v) for v in vals]
    r, triples = cases[0][0], cases[0][2]
    iv = max(max(t) for t in triples) + 2                            # allocation order: count, (x, y, m)*, t, iv, ...
    supplied = [0] + [x for x, _, _ in triples] + [iv]
    with pytest.raises(hip.ZkError):
        hip.WitnessPlan(r, supplied)                                 # not in solved order without the hints
    plan = hip.WitnessPlan(r, supplied, inv_hints=[(x, m) for x, _, m in triples], nonzero_hints=[(x, y) for x, y, _ in triples])
    k = len(cases)
    start = np.zeros((k, r.V + 1, 4), dtype=np.uint64)
    for p in range(k):
        start[p, supplied] = F.fr_to_mont([cases[p][1][i] for i in supplied])
    buf = hip.DeviceBuffer(32 * (r.V + 1) * k)
    buf.upload(start)
    assert plan.solve(buf.ptr, k) == 0
    got = buf.download((k, r.V + 1, 4))
    for p in range(k):
        assert np.array_equal(got[p], F.fr_to_mont(cases[p][1]))
    pk, _ = hip.keygen(r, seed=33)
    pk_o = oracle.pk_from_parts(pk.parts())
    ctx = hip.ProverContext(pk, r, max_batch=k)
    ctx.submit_batch(None, device_ptr=buf.ptr, k=k)
    parts, _ = ctx.collect_batch(k)
    for p in range(k):
        assert hip.proof_to_json(ctx.prove_combine(parts[p]), got[p][1:2]) == oracle.prove(pk_o, r, got[p])[0]
    ctx.close(); plan.close(); buf.free()


def test_prove_batch_chain_shared_sort_and_odd_batch(hip, oracle):
    """dense queries (A-, B-, L-query on one shared witness sort), a batch that is not a power of two, the asynchronous form"""
    r, _ = R.synthetic_chain((1 << 12) - 2, 1)
    pk_o, _ = oracle.keygen(r, seed=12)
    pk = hip.ProvingKey.from_parts(**pk_o.parts())
    ws = [F.fr_to_mont(R.synthetic_chain((1 << 12) - 2, 1, seed=700 + p)[1]) for p in range(5)]
    expect = [oracle.prove(pk_o, r, w)[0] for w in ws]
    ctx = hip.ProverContext(pk, r, max_batch=8)
    assert ctx.info()["share_B"]
    assert hip.prove_batch(ctx, np.stack(ws)) == expect
    k = ctx.submit_batch(np.stack(ws[1:4]))
    parts, _ = ctx.collect_batch(k)
    assert [hip.proof_to_json(ctx.prove_combine(parts[p]), ws[1 + p][1:2]) for p in range(3)] == expect[1:4]
    with pytest.raises(hip.ZkError):
        hip.prove_batch(ctx, np.stack(ws * 2))                      # 10 > max_batch
    ctx.close()


def test_config2_chain_2pow18_gpu_keygen(hip, oracle, tmp_path):
    """BASELINE config 2: synthetic chain, 2^18 constraints, real key from zk_keygen via the .raw file,
    proof bit-exact vs the CPU oracle on the same key"""
    logm = 18
    r, w = R.synthetic_chain((1 << logm) - 2, 1)
    wm = F.fr_to_mont(w)
    pk_path, vk_path = str(tmp_path / "pk.raw"), str(tmp_path / "vk.json")
    assert hip.stub_genkeys_from_pb(r, pk_path, vk_path, seed=18) == 0
    pk = hip.load_proving_key(pk_path)
    pk_o = oracle.read_raw(pk_path)                                          # the oracle reads the file the GPU side wrote
    ctx = hip.ProverContext(pk, r)
    expect, _ = oracle.prove(pk_o, r, wm)
    assert hip.prove(ctx, wm) == expect
    assert expect == oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(18))  # closed form: independent of the .raw codec too


@pytest.mark.parametrize("g2", [False, True], ids=["G1", "G2"])
def test_msm_frugal_table_planes_vs_oracle(hip, oracle, monkeypatch, g2):
    """the memory-frugal table layout at kernel level (every 2^plog-th window tabulated, 2^plog bucket planes folded on the host)"""
    for plog, n, c in [(1, 1000, 0), (2, 20000, 0), (1, 1 << 16, 0), (3, 5000, 11)]:
        monkeypatch.setenv("ZK_TEST_PLANES_LOG", str(plog))
        sc = rand_scalars(n, n + 5, ones_every=9, zeros_every=11)
        bases = tiled_bases(oracle, n, g2=g2)
        s = F.fr_to_mont(sc)
        assert np.array_equal(hip.msm(bases, s, g2=g2, c=c), oracle.msm(bases, s, g2=g2)), (plog, n, c)


def test_prove_2pow18_with_a_quarter_of_the_table_memory(hip, oracle, monkeypatch):
    """a key whose W-fold window-multiple tables do not fit (the reference's domain goes up to 2^28, src/stubs.cpp:49-75): with the table
    budget forced to a quarter of the full tables zk_ctx_create tabulates every 4th window and proves with four bucket planes instead of
    failing; the proof equals the oracle's and the closed form, one at a time and pipelined through the staged pinned-witness path"""
    logm = 18
    r, w = R.synthetic_chain((1 << logm) - 2, 1)
    wm = F.fr_to_mont(w)
    pk, _ = hip.keygen(r, seed=R.SEED_DEFAULT)
    full = hip.ProverContext(pk, r)
    fi = full.info()
    expect = hip.prove(full, wm)
    full.close()
    assert fi["planes"] == 1 and expect == oracle.prove(oracle.pk_from_parts(pk.parts()), r, wm)[0]
    assert expect == oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(R.SEED_DEFAULT))
    parts = pk.parts()
    pk.close()                                                   # drops the full tables (they belong to the key)
    monkeypatch.setenv("ZK_TABLE_BUDGET", str(fi["full_table_bytes"] // 4))
    pk2 = hip.ProvingKey.from_parts(**parts)
    ctx = hip.ProverContext(pk2, r)
    info = ctx.info()
    assert info["planes"] == 4 and info["table_bytes"] <= fi["full_table_bytes"] // 4 and info["table_rows_B"] == -(-info["B"]["W"] // 4)
    assert hip.prove(ctx, wm) == expect
    other = hip.ProverContext(pk2, r)                            # shares the frugal tables; two proofs in flight
    assert other.info()["planes"] == 4
    ctx.submit(wm); other.submit(wm)
    for c in (ctx, other):
        part, _ = c.collect()
        assert hip.proof_to_json(c.prove_combine(part), wm[1:2]) == expect
    ctx.close(); other.close(); pk2.close()


def test_staged_upload_double_buffering(hip, oracle):
    """zk_prove_stage / zk_prove_submit_staged: the next witness is copied while a proof is in flight; proofs unchanged"""
    r, w = R.random_r1cs(900, 2, seed=41)
    r2, w2 = R.random_r1cs(900, 2, seed=41, witness_seed=7)
    wa, wb = F.fr_to_mont(w), F.fr_to_mont(w2)
    pk_o, _ = oracle.keygen(r, seed=6)
    ea, eb = oracle.prove(pk_o, r, wa)[0], oracle.prove(pk_o, r, wb)[0]
    pk = hip.ProvingKey.from_parts(**pk_o.parts())
    c = hip.ProverContext(pk, r, max_batch=2)
    with pytest.raises(hip.ZkError):
        c.submit_staged()                                                    # nothing staged
    c.submit(wa)
    c.stage(wb)                                                              # while proof a is in flight
    with pytest.raises(hip.ZkError):
        c.stage(wb)                                                          # one staged witness per context
    with pytest.raises(hip.ZkError):
        c.submit_staged()                                                    # a is still in flight
    part, _ = c.collect()
    assert hip.proof_to_json(c.prove_combine(part), wa[1:3]) == ea
    c.submit_staged()
    c.stage(np.stack([wa, wb]))                                              # next: a batch of two, staged while b runs
    part, _ = c.collect()
    assert hip.proof_to_json(c.prove_combine(part), wb[1:3]) == eb
    c.submit_staged()
    parts, _ = c.collect_batch(2)
    assert [hip.proof_to_json(c.prove_combine(parts[i:i + 1]), x[1:3]) for i, x in enumerate((wa, wb))] == [ea, eb]
    assert hip.prove(c, wa) == ea                                            # ordinary submit after the buffers were swapped


def test_pinned_witness_submit_and_stage(hip, oracle):
    """SURVEY 8(d)'s hand-over: the witness lies in pinned host memory (zk_host_alloc) and is copied from where it lies --
    zk_prove_submit_pinned on the proof's own stream, zk_prove_stage_pinned ahead of time on the copy stream; a registered
    caller-owned buffer (zk_host_register) works the same; proofs unchanged"""
    r, w = R.random_r1cs(900, 2, seed=41)
    _, w2 = R.random_r1cs(900, 2, seed=41, witness_seed=7)
    wa, wb = F.fr_to_mont(w), F.fr_to_mont(w2)
    pk_o, _ = oracle.keygen(r, seed=6)
    ea, eb = oracle.prove(pk_o, r, wa)[0], oracle.prove(pk_o, r, wb)[0]
    pk = hip.ProvingKey.from_parts(**pk_o.parts())
    c = hip.ProverContext(pk, r, max_batch=2)
    pa, pb = hip.PinnedBuffer(wa.nbytes), hip.PinnedBuffer(2 * wa.nbytes)
    pa.array[:] = wa.reshape(-1)
    pb.array[:] = np.stack([wb, wa]).reshape(-1)
    c.submit_pinned(pa)
    c.stage_pinned(pb, k=2)                                                  # a batch of two, copied while proof a runs
    part, _ = c.collect()
    assert hip.proof_to_json(c.prove_combine(part), wa[1:3]) == ea
    c.submit_staged()
    parts, _ = c.collect_batch(2)
    assert [hip.proof_to_json(c.prove_combine(parts[i:i + 1]), x[1:3]) for i, x in enumerate((wb, wa))] == [eb, ea]
    c.submit_pinned(pb, k=2)
    parts, _ = c.collect_batch(2)
    assert hip.proof_to_json(c.prove_combine(parts[0:1]), wb[1:3]) == eb
    own = np.ascontiguousarray(wb).copy()                                     # a buffer the caller owns, pinned in place
    lib = hip.load_library(hip._lib_path_loaded)
    assert lib.zk_host_register(own.ctypes.data_as(hip.C.c_void_p), hip.C.c_size_t(own.nbytes)) == 0
    c.submit_pinned(own)
    part, _ = c.collect()
    assert hip.proof_to_json(c.prove_combine(part), wb[1:3]) == eb
    assert lib.zk_host_unregister(own.ctypes.data_as(hip.C.c_void_p)) == 0
    pa.free(); pb.free(); c.close()


def test_async_submit_collect_two_contexts(hip, oracle):
    r, w = R.synthetic_chain((1 << 12) - 2, 1)
    wm = F.fr_to_mont(w)
    pk_o, _ = oracle.keygen(r, seed=5)
    expect, _ = oracle.prove(pk_o, r, wm)
    pk = hip.ProvingKey.from_parts(**pk_o.parts())
    a, b = hip.ProverContext(pk, r), hip.ProverContext(pk, r)
    a.submit(wm); b.submit(wm)
    with pytest.raises(hip.ZkError):
        a.submit(wm)                                                         # one proof in flight per context
    for c in (a, b):
        part, _ = c.collect()
        assert hip.proof_to_json(c.prove_combine(part), wm[1:2]) == expect


def test_concurrent_provers_one_context_per_thread(hip, oracle):
    """The reference's threading rule (ProverContext scratch is per prover, the key is shared read-only): four host
    threads, each with its own context on the same key and different witnesses, prove at the same time; contexts are
    created concurrently too (the device-table cache is shared between them)."""
    import threading
    r, _ = R.synthetic_chain((1 << 13) - 2, 1)
    pk_o, _ = oracle.keygen(r, seed=21)
    pk = hip.ProvingKey.from_parts(**pk_o.parts())
    cases = []
    for t in range(4):
        _, w = R.synthetic_chain((1 << 13) - 2, 1, seed=1000 + t)      # same constraint system, different witness
        wm = F.fr_to_mont(w)
        cases.append((wm, oracle.prove(pk_o, r, wm)[0]))
    results, errors = [None] * 4, []

    def worker(t):
        try:
            ctx = hip.ProverContext(pk, r)
            out = [hip.prove(ctx, cases[t][0]) for _ in range(5)]
            ctx.close()
            results[t] = out
        except Exception as e:                                         # surfaced in the main thread below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors
    for t in range(4):
        assert results[t] == [cases[t][1]] * 5


def test_config3_chain_2pow20_headline(hip, oracle):
    """BASELINE config 3 (the bench workload): 2^20-constraint chain, key from zk_keygen; the proof is
    byte-identical to the CPU oracle, and the 4-way base-range sharded path reproduces it"""
    logm = 20
    r, w = R.synthetic_chain((1 << logm) - 2, 1)
    wm = F.fr_to_mont(w)
    pk, _ = hip.keygen(r, seed=R.SEED_DEFAULT)
    ctx = hip.ProverContext(pk, r)
    got = hip.prove(ctx, wm)
    ctx.close()
    expect, _ = oracle.prove(oracle.pk_from_parts(pk.parts()), r, wm)
    assert got == expect
    assert got == oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(R.SEED_DEFAULT))   # closed form from the toxic waste (tcc:533-540)
    shards = []
    for k in range(4):
        c = hip.ProverContext(pk, r, shard_rank=k, shard_count=4)
        shards.append(c.prove_partial(wm))
        c.close()
    ctx = hip.ProverContext(pk, r, shard_rank=0, shard_count=4)
    assert hip.proof_to_json(ctx.prove_combine(np.stack(shards)), wm[1:2]) == expect


def test_config5_size_2pow22_sharded_over_eight(hip, oracle):
    """BASELINE config 5's size (2^22 constraints, the reference's largest named case) on the one GPU of the test box: the
    key comes from zk_keygen and passes the product's own verifier through the proof, and the 8-way base-range sharding
    north_star describes (one shard per GPU, 640-byte partials folded in rank order) reproduces the unsharded proof byte for
    byte -- the eight shards run one after the other here.  The CPU oracle proves the same key and witness at full size too
    (about 15 s on the box's 16 cores): byte-identical JSON."""
    logm = 22
    r, w = R.synthetic_chain((1 << logm) - 2, 1)
    wm = F.fr_to_mont(w)
    pk, vk = hip.keygen(r, seed=R.SEED_DEFAULT)
    ctx = hip.ProverContext(pk, r)
    got = hip.prove(ctx, wm)
    ctx.close()
    assert hip.stub_verify(vk.to_json(), got)
    expect, _ = oracle.prove(oracle.pk_from_parts(pk.parts()), r, wm)        # the oracle at 2^22
    assert got == expect
    assert got == oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(R.SEED_DEFAULT))   # closed form: O(nnz) field work + 3 scalar multiplications
    d = json.loads(got)
    d["input"][0] = d["input"][0][:-1] + ("0" if d["input"][0][-1] != "0" else "1")      # another public input
    assert not hip.stub_verify(vk.to_json(), json.dumps(d))
    shards = []
    for k in range(8):
        c = hip.ProverContext(pk, r, shard_rank=k, shard_count=8)
        shards.append(c.prove_partial(wm))
        c.close()
    ctx = hip.ProverContext(pk, r, shard_rank=0, shard_count=8)
    assert hip.proof_to_json(ctx.prove_combine(np.stack(shards)), wm[1:2]) == got
    ctx.close()


@pytest.mark.parametrize("logm", [12, 18])
def test_option2_split_witness_map_entry_points_on_one_device(hip, oracle, logm):
    """SURVEY 8(e) option 2 on real hardware as far as one GPU allows: three same-device contexts own shards 0..2 of 3, context k
    runs transform chain k (zk_chain_submit: A, B, C), context 0 forms H from the three chain buffers handed over as plain device
    pointers (zk_h_from_chains_submit), every context proves its shard from h[lo..hi) (zk_prove_submit_with_h); the folded proof
    equals the oracle's and the closed form; an unsatisfying witness fails the degree check where H is formed."""
    r, w = R.synthetic_chain((1 << logm) - 2, 1)
    wm = F.fr_to_mont(w)
    pk, _ = hip.keygen(r, seed=70 + logm)
    expect, _ = oracle.prove(oracle.pk_from_parts(pk.parts()), r, wm)
    assert expect == oracle.proof_from_trapdoor(r, wm, oracle.toxic_from_seed(70 + logm))
    G, m = 3, r.domain_size
    ctxs = [hip.ProverContext(pk, r, shard_rank=k, shard_count=G) for k in range(G)]

    def run(wit):
        for k in range(G):
            ctxs[k].chain_submit(wit, k)
        for k in range(G):
            ctxs[k].chain_wait()
        ctxs[0].h_from_chains_submit(ctxs[0].chain_device_ptr(0), ctxs[1].chain_device_ptr(1), ctxs[2].chain_device_ptr(2))
        ctxs[0].chain_wait(check_degree=True)
        for k in range(G):
            ctxs[k].submit_with_h(wit, ctxs[0].h_device_ptr() + 32 * ((m - 1) * k // G))
        parts = [ctxs[k].collect()[0] for k in range(G)]
        return hip.proof_to_json(ctxs[0].prove_combine(np.stack(parts)), wit[1:2])

    def run_pipelined(wit):
        """the two-step form the multi-rank driver uses: the witness part of every shard is queued before H exists"""
        for k in range(G):
            ctxs[k].submit_defer_h(wit)
        try:
            for k in range(G):
                ctxs[k].chain_submit(None, k)                                # witness = the deferred proof's
            for k in range(G):
                ctxs[k].chain_wait()
            ctxs[0].h_from_chains_submit(ctxs[0].chain_device_ptr(0), ctxs[1].chain_device_ptr(1), ctxs[2].chain_device_ptr(2))
            ctxs[0].chain_wait(check_degree=True)
        except hip.ZkError:
            for k in range(G):
                ctxs[k].abort()
            raise
        with pytest.raises(hip.ZkError):
            ctxs[1].collect()                                                # no H part yet
        for k in range(G):
            ctxs[k].submit_h(ctxs[0].h_device_ptr() + 32 * ((m - 1) * k // G))
        parts = [ctxs[k].collect()[0] for k in range(G)]
        return hip.proof_to_json(ctxs[0].prove_combine(np.stack(parts)), wit[1:2])

    assert run(wm) == expect
    assert run(wm) == expect                                                 # the contexts are reusable
    assert run_pipelined(wm) == expect
    with pytest.raises(hip.ZkError):
        ctxs[0].submit_h(ctxs[0].h_device_ptr())                             # nothing deferred
    bad = wm.copy(); bad[5] = wm[6]
    with pytest.raises(hip.ZkError) as e:
        run(bad)
    assert e.value.code == 7                                                 # ZK_ERR_DEGREE, raised where H is formed
    assert run(wm) == expect                                                 # ... and the contexts survive it
    with pytest.raises(hip.ZkError) as e:
        run_pipelined(bad)
    assert e.value.code == 7
    assert run_pipelined(wm) == expect
    # the replicated path of the same shards gives the same proof
    parts = [c.prove_partial(wm) for c in ctxs]
    assert hip.proof_to_json(ctxs[0].prove_combine(np.stack(parts)), wm[1:2]) == expect
    for c in ctxs:
        c.close()


def test_assembly_postpass_off_and_on_give_identical_results(hip, oracle):
    """The default build passes the device assembly through tools/strip_asm_nops.py (drops the s_nop hipcc pads behind the
    inline-asm statements of the field arithmetic).  The same sources built with POSTPASS=0 (variants/nopostpass, made by
    __graft_entry__.build()) must give the same bytes: a G1 and a G2 multi-exponentiation and a proof, in a child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "variants", "nopostpass", "libzkhip.so")
    assert os.path.exists(so), "variants/nopostpass/libzkhip.so missing: run __graft_entry__.build()"
    code = '''
import sys, json, numpy as np
sys.path[:0] = [%r, %r, %r]
import oracle_lib as O
from helpers import rand_scalars, tiled_bases
from ethsnarks_amd import prover as P, r1cs as R, fields as F
P.load_library(%r)
out = {}
for g2 in (False, True):
    s = F.fr_to_mont(rand_scalars(5000, 3, ones_every=7, zeros_every=13))
    out["msm_g2" if g2 else "msm_g1"] = P.msm(tiled_bases(O, 5000, g2=g2), s, g2=g2).tolist()
r, w = R.synthetic_chain((1 << 12) - 2, 1)
pk, _ = P.keygen(r, seed=4)
out["proof"] = P.prove(P.ProverContext(pk, r), F.fr_to_mont(w))
print("RESULT" + json.dumps(out))
'''
    outs = []
    for lib in (so, os.path.join(root, "ethsnarks_amd", "libzkhip.so")):
        p = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests"), os.path.join(root, "oracle"), lib)], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append([l for l in p.stdout.splitlines() if l.startswith("RESULT")][0])
    assert outs[0] == outs[1]
    r, w = R.synthetic_chain((1 << 12) - 2, 1)
    pk_o, _ = oracle.keygen(r, seed=4)
    assert json.loads(outs[0][6:])["proof"] == oracle.prove(pk_o, r, F.fr_to_mont(w))[0]


def test_stub_test_proof_verify_and_static_triple(hip):
    """stub_test_proof_verify (src/stubs.cpp:135-148) through the C ABI on the GPU; zk_verify on the reference's vector"""
    import os
    from helpers import GOLDEN
    r, w = R.random_r1cs(200, 3, seed=21)
    assert hip.stub_test_proof_verify(r, F.fr_to_mont(w), seed=2)
    d = json.load(open(os.path.join(GOLDEN, "ref_static_triple.json")))
    assert hip.stub_verify(json.dumps(d["vk"]), json.dumps(d["proof"]))
    assert not hip.stub_verify(json.dumps(d["vk"]), json.dumps(dict(d["proof"], C=d["proof"]["A"])))


def test_sharded_device_exchange_over_rccl_one_rank(hip):
    """The device-buffer exchange of sharded provers on real hardware as far as one GPU allows (tests/rccl_one_rank.py, run
    as a child process so that torch initialises the GPU before the library does, as in bench.py): a context that owns shard
    0 of 2 leaves its partial sums in its 640-byte device buffer, torch wraps that buffer without a copy, RCCL ("nccl", world
    size 1) all-gathers it, zk_prove_combine_device folds the gathered record; the two shards folded together give the
    oracle's proof."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "rccl_one_rank.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
