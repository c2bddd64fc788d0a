"""Kernel-logic tests without a GPU: the HIP sources compiled against tests/emul/hip_emul.h
(libzkhip_emul.so, CPU stand-in, test infrastructure only) are checked against the oracle.
The same checks run on the real device in test_gpu_parity.py."""
import json
import numpy as np
import pytest
from ethsnarks_amd import r1cs as R, fields as F
from helpers import golden_cases, build_case, rand_scalars


@pytest.fixture(scope="module")
def zk(emul):
    from ethsnarks_amd import prover
    prover._lib = None
    prover._lib_path_loaded = None
    prover.load_library(emul)
    assert b"EMULATION" in prover._lib.zk_version()
    yield prover
    prover._lib = None
    prover._lib_path_loaded = None


def test_field_mul(zk, oracle):
    a = F.fr_to_mont(rand_scalars(70, 1)); b = F.fr_to_mont(rand_scalars(70, 2))
    o = np.zeros_like(a)
    oracle.lib().orc_fr_mul(oracle._p64(o), oracle._p64(a), oracle._p64(b), 70)
    assert np.array_equal(zk.field_mul(a, b, "fr"), o)
    oracle.lib().orc_fq_mul(oracle._p64(o), oracle._p64(a), oracle._p64(b), 70)
    assert np.array_equal(zk.field_mul(a, b, "fq"), o)


@pytest.mark.parametrize("logm", [0, 1, 2, 3, 5, 11, 12, 13, 14])      # stage counts per pass: odd and even, one and several pairs
def test_ntt(zk, oracle, logm):
    x = F.fr_to_mont(rand_scalars(1 << logm, logm))
    for inv in (False, True):
        for coset in (False, True):
            assert np.array_equal(zk.ntt(x, logm, inv, coset), oracle.ntt(x, logm, inv, coset))


@pytest.mark.parametrize("g2", [False, True])
@pytest.mark.parametrize("n,c", [(0, 0), (1, 0), (7, 0), (300, 4), (1000, 0), (1000, 9)])
def test_msm(zk, oracle, g2, n, c):
    sc = rand_scalars(n, n + 1, ones_every=5, zeros_every=7)
    if n > 20:
        sc[11] = F.FR - 1; sc[12] = 2; sc[13] = 1 << 253
    s = F.fr_to_mont(sc) if n else np.zeros((0, 4), dtype=np.uint64)
    bases = oracle.batch_mul(F.fr_to_mont(rand_scalars(n, 99)), g2=g2) if n else np.zeros((0, 16 if g2 else 8), dtype=np.uint64)
    if n > 30:
        bases[20] = 0; bases[21] = bases[22]
    assert np.array_equal(zk.msm(bases, s, g2=g2, c=c), oracle.msm(bases, s, g2=g2))


def test_msm_heavy_bucket(zk, oracle):
    """thousands of equal scalars (the scalar==1 partition of *_with_mixed_addition): workgroup path"""
    n = 3600
    s = F.fr_to_mont([1] * 3000 + [77] * 500 + rand_scalars(100, 5))
    bases = oracle.batch_mul(F.fr_to_mont(rand_scalars(n, 6)))
    assert np.array_equal(zk.msm(bases, s, c=6), oracle.msm(bases, s))


@pytest.mark.parametrize("quad,quad_acc", [("0", "0"), ("1", "0"), ("0", "1")])
def test_msm_lane_layouts(zk, oracle, monkeypatch, quad, quad_acc):
    """small MSMs default to four lanes per logical thread (Curve::*_q); the one-lane kernels of the big sizes and the
    mixed layouts must index the same way (ZK_MSM_QUAD / ZK_MSM_QUAD_ACC override the size rule)"""
    monkeypatch.setenv("ZK_MSM_QUAD", quad); monkeypatch.setenv("ZK_MSM_QUAD_ACC", quad_acc)
    n = 700
    s = F.fr_to_mont([1] * 300 + rand_scalars(n - 300, 8, zeros_every=9))
    for g2 in (False, True):
        bases = oracle.batch_mul(F.fr_to_mont(rand_scalars(n, 21)), g2=g2)
        assert np.array_equal(zk.msm(bases, s, g2=g2, c=5), oracle.msm(bases, s, g2=g2))


def test_msm_all_cancel(zk, oracle):
    """P and -P with equal scalars, and P + P: exceptional cases of the mixed addition"""
    pts = oracle.batch_mul(F.fr_to_mont([5, 5, 9]))
    neg = pts[0].copy()
    y = F.fq_from_mont(neg[4:8].reshape(1, 4))[0]
    neg[4:8] = F.fq_to_mont([(-y) % F.FQ])[0]
    bases = np.stack([pts[0], neg, pts[2], pts[2]])
    s = F.fr_to_mont([3, 3, 4, 4])
    got = zk.msm(bases, s, c=3)
    assert np.array_equal(got, oracle.msm(bases, s, naive=True))
    s0 = F.fr_to_mont([3, 3, 0, 0])
    assert (zk.msm(bases, s0, c=3) == 0).all()          # infinity


@pytest.mark.parametrize("case", golden_cases()[:3], ids=lambda c: c["name"])
def test_prove_golden(zk, oracle, case):
    r, w, toxic = build_case(case)
    pk_o, _ = oracle.keygen(r, toxic=toxic)
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    ctx = zk.ProverContext(pk, r)
    assert zk.prove(ctx, F.fr_to_mont(w)) == case["proof_json"]
    assert zk.prove(ctx, F.ints_to_limbs(w), canonical=True) == case["proof_json"]
    ctx.close()
    one = zk.ProverContext(pk, r, one_stream=True)          # zk_config.schedule = ZK_SCHED_ONE_STREAM
    assert zk.prove(one, F.fr_to_mont(w)) == case["proof_json"]
    one.close()


def test_prove_shape_sweep(zk, oracle):
    """domain-size boundaries, no public inputs, a single constraint (the GPU suite sweeps 26 shapes up to 2^11)"""
    for i, (nC, nIn) in enumerate([(1, 0), (3, 2), (7, 0), (62, 1), (63, 0)]):
        r, w = R.random_r1cs(nC, nIn, n_extra_vars=i % 4, max_terms=1 + i % 8, seed=1000 + i, small_values=(i % 3 == 1))
        wm = F.fr_to_mont(w)
        pk_o, _ = oracle.keygen(r, seed=77 + i)
        pk = zk.ProvingKey.from_parts(**pk_o.parts())
        ctx = zk.ProverContext(pk, r, one_stream=bool(i & 1))
        assert zk.prove(ctx, wm) == oracle.prove(pk_o, r, wm)[0], (nC, nIn)
        ctx.close(); pk.close()


def test_staged_upload(zk, oracle):
    """zk_prove_stage / zk_prove_submit_staged (double-buffered upload): same proofs, buffers swap cleanly"""
    r, w = R.random_r1cs(40, 1, seed=3)
    _, w2 = R.random_r1cs(40, 1, seed=3, witness_seed=9)
    wa, wb = F.fr_to_mont(w), F.fr_to_mont(w2)
    pk_o, _ = oracle.keygen(r, seed=2)
    ea, eb = oracle.prove(pk_o, r, wa)[0], oracle.prove(pk_o, r, wb)[0]
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    c = zk.ProverContext(pk, r)
    with pytest.raises(zk.ZkError):
        c.submit_staged()
    c.submit(wa); c.stage(wb)
    with pytest.raises(zk.ZkError):
        c.submit_staged()                                                    # a is in flight
    part, _ = c.collect()
    assert zk.proof_to_json(c.prove_combine(part), wa[1:2]) == ea
    c.submit_staged()
    part, _ = c.collect()
    assert zk.proof_to_json(c.prove_combine(part), wb[1:2]) == eb
    assert zk.prove(c, wa) == ea
    c.close(); pk.close()


def test_pinned_witness_and_two_phase_submit(zk, oracle):
    """round-3 entry points in the emulation build: the witness read from a zk_host_alloc buffer where it lies (submit / stage), and
    the two-step submit of sharded latency mode (zk_prove_submit_defer_h -> zk_chain_submit(NULL) -> zk_h_from_chains_submit ->
    zk_prove_submit_h) on ONE unsharded context, with its state machine: collect refuses a proof without its H part, abort frees it"""
    r, w = R.random_r1cs(300, 2, seed=15)
    wm = F.fr_to_mont(w)
    pk_o, _ = oracle.keygen(r, seed=9)
    expect = oracle.prove(pk_o, r, wm)[0]
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    c = zk.ProverContext(pk, r)
    pin = zk.PinnedBuffer(wm.nbytes); pin.array[:] = wm.reshape(-1)
    c.submit_pinned(pin); c.stage_pinned(pin)
    part, _ = c.collect()
    assert zk.proof_to_json(c.prove_combine(part), wm[1:3]) == expect
    c.submit_staged()
    part, _ = c.collect()
    assert zk.proof_to_json(c.prove_combine(part), wm[1:3]) == expect
    with pytest.raises(zk.ZkError):
        c.submit_h(c.h_device_ptr())                               # nothing deferred
    c.submit_defer_h(wm)
    with pytest.raises(zk.ZkError):
        c.collect()                                                # no H part yet
    with pytest.raises(zk.ZkError):
        c.submit(wm)                                               # a proof is in flight
    for which in range(3):
        c.chain_submit(None, which)                                # the deferred proof's witness
    c.chain_wait()
    c.h_from_chains_submit(c.chain_device_ptr(0), c.chain_device_ptr(1), c.chain_device_ptr(2))
    c.chain_wait(check_degree=True)
    c.submit_h(c.h_device_ptr())
    part, _ = c.collect()
    assert zk.proof_to_json(c.prove_combine(part), wm[1:3]) == expect
    c.submit_defer_h(wm); c.abort()                                # dropped: the context is free again
    assert zk.prove(c, wm) == expect
    pin.free(); c.close()


def test_prove_chain_long_rows_and_sharding(zk, oracle, tmp_path):
    r, w = R.synthetic_chain(254, 1)                    # last row: 510 terms -> long-row path
    wm = F.fr_to_mont(w)
    pk_o, _ = oracle.keygen(r, seed=3)
    expect, _ = oracle.prove(pk_o, r, wm)
    path = str(tmp_path / "pk.raw")
    pk_o.write_raw(path)
    pk = zk.load_proving_key(path)
    ctx = zk.ProverContext(pk, r)
    assert np.array_equal(ctx.witness_map(wm), oracle.witness_map(r, wm))
    assert zk.prove(ctx, wm) == expect
    assert zk.stub_prove_from_pb(r, wm, path) == expect
    parts = [zk.ProverContext(pk, r, shard_rank=k, shard_count=3).prove_partial(wm) for k in range(3)]
    assert zk.proof_to_json(ctx.prove_combine(np.stack(parts)), wm[1:2]) == expect
    out = str(tmp_path / "pk2.raw")
    pk.save_raw(out)
    assert open(out, "rb").read() == open(path, "rb").read()


def test_errors(zk, oracle, tmp_path):
    r, w = R.random_r1cs(12, 1, seed=2)
    pk_o, _ = oracle.keygen(r, seed=3)
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    ctx = zk.ProverContext(pk, r)
    bad = list(w); bad[-4] = (bad[-4] + 1) % F.FR
    with pytest.raises(zk.ZkError) as e:
        zk.prove(ctx, F.fr_to_mont(bad))
    assert e.value.code == 7                                   # ZK_ERR_DEGREE
    r2, _ = R.random_r1cs(20, 1, seed=2)
    with pytest.raises(zk.ZkError) as e:
        zk.ProverContext(pk, r2)
    assert e.value.code == 6                                   # ZK_ERR_SHAPE
    with pytest.raises(zk.ZkError) as e:
        zk.load_proving_key(str(tmp_path / "nope.raw"))
    assert e.value.code == 2                                   # ZK_ERR_IO
    p = tmp_path / "junk.raw"; p.write_bytes(b"0" + b"\x01" * 40)
    with pytest.raises(zk.ZkError) as e:
        zk.load_proving_key(str(p))
    assert e.value.code == 3                                   # ZK_ERR_FORMAT


def test_verifier_on_reference_static_triple(zk):
    """the product's C++ verifier is pinned directly on the reference's static (vk, proof) vector"""
    import json, os
    from helpers import GOLDEN
    d = json.load(open(os.path.join(GOLDEN, "ref_static_triple.json")))
    assert zk.stub_verify(json.dumps(d["vk"]), json.dumps(d["proof"]))
    bad = dict(d["proof"]); bad["input"] = [bad["input"][0], "0x8"]
    assert not zk.stub_verify(json.dumps(d["vk"]), json.dumps(bad))
    assert not zk.stub_verify(json.dumps(d["vk"]), json.dumps(dict(d["proof"], A=d["proof"]["C"])))
    assert not zk.stub_verify(json.dumps(d["vk"]), json.dumps(dict(d["proof"], input=d["proof"]["input"][:1])))
    with pytest.raises(zk.ZkError):
        zk.stub_verify("{}", json.dumps(d["proof"]))


def test_stub_test_proof_verify_roundtrip(zk):
    """keygen -> prove -> verify entirely through the C ABI (src/stubs.cpp:135-148)"""
    r, w = R.random_r1cs(20, 2, seed=8)
    assert zk.stub_test_proof_verify(r, F.fr_to_mont(w), seed=3)


def test_no_public_inputs_and_tiny_domain(zk, oracle):
    """nIn = 0 (the reference's own Merkle test declares no public input): JSON tail is `"input" :[]`;
    and the smallest circuits (domain 2 and 4)"""
    import json
    import pyref
    for nC in (1, 2, 6):
        r, w = R.random_r1cs(nC, 0, seed=30 + nC)
        wm = F.fr_to_mont(w)
        pk_o, vk_o = oracle.keygen(r, seed=9)
        expect, _ = oracle.prove(pk_o, r, wm)
        assert expect.endswith('"input" :[]\n}')
        ctx = zk.ProverContext(zk.ProvingKey.from_parts(**pk_o.parts()), r)
        got = zk.prove(ctx, wm)
        assert got == expect
        assert zk.stub_verify(vk_o.to_json(), got)
        assert pyref.verify(json.loads(vk_o.to_json()), json.loads(got))


@pytest.mark.parametrize("k", [1, 3])
def test_prove_batch_equals_independent_proofs(zk, oracle, k):
    """zk_prove_batch: k witnesses of one circuit through one launch sequence (sort key (proof, bucket)); proof p must be
    byte-identical to the oracle's proof of witness p.  k = 3: a batch that is not a power of two (partly filled last
    coarse bin of the sort).  Sparse B / A queries and unreferenced variables: the non-shared-sort and gather paths."""
    r, _ = R.random_r1cs(40, 2, n_extra_vars=3, seed=31)
    pk_o, _ = oracle.keygen(r, seed=8)
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    ws, expect = [], []
    for p in range(k):
        r_p, w = R.random_r1cs(40, 2, n_extra_vars=3, seed=31, witness_seed=(100 + p) if p else None)
        assert np.array_equal(r_p.A.col, r.A.col) and np.array_equal(r_p.C.coeff, r.C.coeff)      # same constraint system
        wm = F.fr_to_mont(w)
        ws.append(wm); expect.append(oracle.prove(pk_o, r, wm)[0])
    ctx = zk.ProverContext(pk, r, max_batch=4)
    assert zk.prove_batch(ctx, np.stack(ws)) == expect
    assert zk.prove(ctx, ws[-1]) == expect[-1]                       # the same context still proves one at a time
    if k > 1:
        with pytest.raises(zk.ZkError):
            zk.prove_batch(ctx, np.stack(ws * 3))                    # 3k > max_batch
        bad = [w.copy() for w in ws]
        bad[1] = F.fr_to_mont([(int(v) + (i == len(bad[1]) - 4)) % F.FR for i, v in enumerate(F.fr_from_mont(bad[1]))])
        with pytest.raises(zk.ZkError) as e:
            zk.prove_batch(ctx, np.stack(bad))                       # one unsatisfied witness fails the batch (degree check)
        assert e.value.code == 7
    ctx.close()


def test_prove_batch_chain_shared_sort(zk, oracle):
    """dense queries (the chain circuit): the A-, B- and L-query ride ONE shared sort of the k witness digit streams"""
    r, w0 = R.synthetic_chain(254, 1)
    pk_o, _ = oracle.keygen(r, seed=4)
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    ws = [F.fr_to_mont(R.synthetic_chain(254, 1, seed=500 + p)[1]) for p in range(3)]
    expect = [oracle.prove(pk_o, r, w)[0] for w in ws]
    ctx = zk.ProverContext(pk, r, max_batch=3)
    assert ctx.info()["share_B"]
    assert zk.prove_batch(ctx, np.stack(ws)) == expect
    k = ctx.submit_batch(np.stack(ws[:2]))                            # asynchronous form, a smaller batch than the capacity
    parts, _ = ctx.collect_batch(k)
    assert [zk.proof_to_json(ctx.prove_combine(parts[p]), ws[p][1:2]) for p in range(2)] == expect[:2]
    ctx.close()


def test_witness_plan_completes_witnesses_on_the_device(zk, oracle):
    """zk_wplan: the MiMC hash circuit's constraint system, compiled as a forward-substitution program, fills in
    k witnesses from their supplied variables only (digest, IV, message words) -- equal to the front end's witnesses -- and the
    buffer goes straight to zk_prove_batch_submit_resident.  A wrong digest is reported as a violated constraint; a system
    that is not in solved order is refused with an explanation."""
    from ethsnarks_amd import gadgets as G
    k = 3
    r, w0, _ = G.mimc_preimage_circuit(2, seed=7)
    full = [w0] + [G.mimc_preimage_circuit(2, seed=70 + p)[1] for p in range(1, k)]
    supplied = list(range(0, 1 + 1 + 1 + 2))                       # ONE, digest, iv, m[0..1] (allocation order of the circuit)
    plan = zk.WitnessPlan(r, supplied)
    buf = zk.DeviceBuffer(32 * (r.V + 1) * k)
    start = np.zeros((k, r.V + 1, 4), dtype=np.uint64)
    for p in range(k):
        start[p, supplied] = F.fr_to_mont([full[p][i] for i in supplied])
    buf.upload(start)
    assert plan.solve(buf.ptr, k) == 0
    got = buf.download((k, r.V + 1, 4))
    for p in range(k):
        assert np.array_equal(got[p], F.fr_to_mont(full[p]))
    pk_o, _ = oracle.keygen(r, seed=5)
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    ctx = zk.ProverContext(pk, r, max_batch=k)
    ctx.submit_batch(None, device_ptr=buf.ptr, k=k)
    parts, _ = ctx.collect_batch(k)
    for p in range(k):
        assert zk.proof_to_json(ctx.prove_combine(parts[p]), got[p][1:2]) == oracle.prove(pk_o, r, got[p])[0]
    start[1, 1] = F.fr_to_mont([12345])[0]                          # a wrong digest: the final check constraint does not hold
    buf.upload(start)
    assert plan.solve(buf.ptr, k) == 1
    with pytest.raises(zk.ZkError) as e:
        zk.WitnessPlan(r, supplied[:-1])                            # a message word neither supplied nor defined
    assert e.value.code == 1 and "solved order" in str(e.value)
    ctx.close(); plan.close(); buf.free()


def test_witness_plan_with_bit_decomposition_hints(zk, oracle):
    """Gadgets with non-deterministic advice (the reference's field2bits family): the constraints only CHECK the bits, so the forward
    substitution needs a hint -- ZK_WHINT_BITS: w[first + i] = bit i of w[src].  With it the plan completes the witness from x alone
    (equal to the front end's, proof equal to the oracle's); without it the system is refused."""
    from ethsnarks_amd import gadgets as G
    k = 3
    cases = [G.field2bits_circuit(64, seed=40 + p) for p in range(k)]
    r, (xv, first, nb) = cases[0][0], cases[0][2]
    iv = first + nb + 1                                              # allocation order: x, bits, y, iv, ...
    supplied = [0, xv, iv]
    with pytest.raises(zk.ZkError) as e:
        zk.WitnessPlan(r, supplied)
    assert "solved order" in str(e.value)
    plan = zk.WitnessPlan(r, supplied, bit_hints=[(xv, first, nb)])
    start = np.zeros((k, r.V + 1, 4), dtype=np.uint64)
    for p in range(k):
        start[p, supplied] = F.fr_to_mont([cases[p][1][i] for i in supplied])
    buf = zk.DeviceBuffer(32 * (r.V + 1) * k)
    buf.upload(start)
    assert plan.solve(buf.ptr, k) == 0
    got = buf.download((k, r.V + 1, 4))
    for p in range(k):
        assert np.array_equal(got[p], F.fr_to_mont(cases[p][1]))
    pk_o, _ = oracle.keygen(r, seed=8)
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    ctx = zk.ProverContext(pk, r, max_batch=k)
    ctx.submit_batch(None, device_ptr=buf.ptr, k=k)
    parts, _ = ctx.collect_batch(k)
    for p in range(k):
        assert zk.proof_to_json(ctx.prove_combine(parts[p]), got[p][1:2]) == oracle.prove(pk_o, r, got[p])[0]
    ctx.close(); plan.close(); buf.free()


def test_witness_plan_with_inverse_and_nonzero_hints(zk, oracle):
    """The reference's IsNonZero gadget (src/gadgets/isnonzero.cpp): M = 1 / X (0 for 0) and Y = [X != 0] are advice that its three
    constraints only check, and the first of them reads Y before anything defines it.  ZK_WHINT_INV / ZK_WHINT_NONZERO supply both: the plan
    completes the witness from the X alone (zero and non-zero values), equal to the front end's; a wrong hint set is refused or caught."""
    from ethsnarks_amd import gadgets as G
    vals = [(0, 5, 0, 7, 1), (3, 0, 0, 0, 9), (0, 0, 0, 0, 0)]
    cases = [G.isnonzero_circuit(v) for v in vals]
    r, triples = cases[0][0], cases[0][2]
    iv = max(max(t) for t in triples) + 2                            # allocation order: count, (x, y, m)*, t, iv, ...
    supplied = [0] + [x for x, _, _ in triples] + [iv]
    with pytest.raises(zk.ZkError) as e:
        zk.WitnessPlan(r, supplied)
    assert "solved order" in str(e.value)
    plan = zk.WitnessPlan(r, supplied, inv_hints=[(x, m) for x, _, m in triples], nonzero_hints=[(x, y) for x, y, _ in triples])
    k = len(cases)
    start = np.zeros((k, r.V + 1, 4), dtype=np.uint64)
    for p in range(k):
        start[p, supplied] = F.fr_to_mont([cases[p][1][i] for i in supplied])
    buf = zk.DeviceBuffer(32 * (r.V + 1) * k)
    buf.upload(start)
    assert plan.solve(buf.ptr, k) == 0
    got = buf.download((k, r.V + 1, 4))
    for p in range(k):
        assert np.array_equal(got[p], F.fr_to_mont(cases[p][1]))
    plan.close()
    # the hints exchanged (Y from the inverse, M from the flag): the checks catch it wherever X is neither 0 nor 1
    bad = zk.WitnessPlan(r, supplied, inv_hints=[(x, y) for x, y, _ in triples], nonzero_hints=[(x, m) for x, _, m in triples])
    buf.upload(start)
    assert bad.solve(buf.ptr, k) > 0
    bad.close(); buf.free()



@pytest.mark.parametrize("plog", [1, 2])
def test_msm_frugal_tables_planes(zk, oracle, monkeypatch, plog):
    """memory-frugal table layout at kernel level (MsmShape::plog): every 2^plog-th window tabulated, 2^plog bucket planes, planes
    folded on the host -- the same group element as the full tables (ZK_TEST_PLANES_LOG is a test aid of zk_msm_g1 / zk_msm_g2)"""
    monkeypatch.setenv("ZK_TEST_PLANES_LOG", str(plog))
    for g2 in (False, True):
        for n, c in [(1, 0), (7, 0), (300, 4), (1000, 9)]:
            sc = rand_scalars(n, n + 3, ones_every=5, zeros_every=7)
            if n > 20:
                sc[11] = F.FR - 1; sc[12] = 2; sc[13] = 1 << 253
            bases = oracle.batch_mul(F.fr_to_mont(rand_scalars(n, 98)), g2=g2)
            if n > 30:
                bases[20] = 0; bases[21] = bases[22]
            s = F.fr_to_mont(sc)
            assert np.array_equal(zk.msm(bases, s, g2=g2, c=c), oracle.msm(bases, s, g2=g2)), (g2, n, c)


@pytest.mark.parametrize("frac", [4, 16])
def test_prove_with_a_table_budget_below_the_full_tables(zk, oracle, monkeypatch, frac):
    """ZK_TABLE_BUDGET below the W-fold tables: zk_ctx_create keeps every S-th window instead of failing (S = planes), the proof bytes do
    not change -- single proofs, a batch, the merged H + L tail, and a sharded context's device-side partial sums"""
    r, w = R.synthetic_chain(254, 1)
    wm = F.fr_to_mont(w)
    pk_o, _ = oracle.keygen(r, seed=31)
    expect = oracle.prove(pk_o, r, wm)[0]
    pk = zk.ProvingKey.from_parts(**pk_o.parts())
    full = zk.ProverContext(pk, r)
    fi = full.info()
    assert fi["planes"] == 1 and fi["table_bytes"] == fi["full_table_bytes"]
    full.close(); pk.close()
    monkeypatch.setenv("ZK_TABLE_BUDGET", str(fi["full_table_bytes"] // frac))
    pk = zk.ProvingKey.from_parts(**pk_o.parts())          # a new key object: the table cache is per key
    ctx = zk.ProverContext(pk, r, max_batch=3)
    info = ctx.info()
    assert info["planes"] >= 2 and info["table_bytes"] <= fi["full_table_bytes"] // frac + 4096 * 320 or info["planes"] == 16
    assert info["table_rows_B"] == -(-info["B"]["W"] // info["planes"])
    assert zk.prove(ctx, wm) == expect
    _, w2 = R.synthetic_chain(254, 1, seed=R.SEED_DEFAULT + 1)
    wm2 = F.fr_to_mont(w2)
    assert zk.prove_batch(ctx, np.stack([wm, wm2, wm])) == [expect, oracle.prove(pk_o, r, wm2)[0], expect]
    ctx.close()
    parts = []
    monkeypatch.setenv("ZK_TABLE_BUDGET", str(fi["full_table_bytes"] // (3 * frac)))      # (a shard holds a third of every table)
    for k in range(3):                                     # sharded contexts: the device copy of the partial sums is written after the host fold
        sc = zk.ProverContext(pk, r, shard_rank=k, shard_count=3)
        assert sc.info()["planes"] >= 2
        sc.submit(wm); sc.collect_device()
        import ctypes as C
        buf = np.zeros(80, dtype=np.uint64)
        assert zk._lib.zk_dev_download(buf.ctypes.data_as(C.c_void_p), C.c_void_p(sc.partials_device_ptr()), C.c_size_t(640)) == 0
        parts.append(buf)
        last = sc
    assert zk.proof_to_json(last.prove_combine(np.stack(parts)), wm[1:2]) == expect
    pk.close()


def test_two_devices_two_keys_table_cache_and_current_device(emul, oracle):
    """the table cache evicted idle tables of ANY device and left the current device switched.  In a child process
    with two emulated devices (ZK_EMUL_DEVICES=2; the emulator tracks the calling thread's current device, the free memory it reports per
    device and the allocations made on each): a context for key 2 on device 1 that is short of memory must not drop key 1's idle tables
    on device 0, everything it allocates must be allocated with device 1 current, and freeing a key whose tables live on device 1 from a
    thread whose current device is 0 must leave device 0 current"""
    import subprocess, sys, os
    from conftest import ROOT
    child = r'''
import ctypes as C, os, sys
sys.path[:0] = [sys.argv[2], os.path.join(sys.argv[2], "tests"), os.path.join(sys.argv[2], "oracle")]
import numpy as np
import oracle_lib as O
from ethsnarks_amd import prover as P, r1cs as R, fields as F
P.load_library(sys.argv[1])
L = P._lib
L.zk_emul_alloc_count.restype = C.c_uint64
allocs = lambda d: int(L.zk_emul_alloc_count(d))
r, w = R.synthetic_chain(126, 1)
wm = F.fr_to_mont(w)
pk_o1, _ = O.keygen(r, seed=1); pk_o2, _ = O.keygen(r, seed=2)
e1, e2 = O.prove(pk_o1, r, wm)[0], O.prove(pk_o2, r, wm)[0]
k1 = P.ProvingKey.from_parts(**pk_o1.parts()); k2 = P.ProvingKey.from_parts(**pk_o2.parts())
a0 = allocs(0)
c = P.ProverContext(k1, r, device=0); assert P.prove(c, wm) == e1; c.close()        # key 1's tables stay cached on device 0 (idle)
fresh = allocs(0) - a0
a0 = allocs(0)
c = P.ProverContext(k1, r, device=0); c.close()
cached = allocs(0) - a0
assert cached < fresh, (cached, fresh)                                             # the second context found the tables
L.zk_emul_set_free_mem(1, C.c_size_t(1 << 20))                                      # device 1 is short of memory: the eviction loop runs
a0, a1 = allocs(0), allocs(1)
c2 = P.ProverContext(k2, r, device=1)
assert allocs(0) == a0, "allocations of a device-1 context landed on device 0"
assert allocs(1) > a1
assert L.zk_emul_current_device() == 1
assert P.prove(c2, wm) == e2
a0 = allocs(0)
c = P.ProverContext(k1, r, device=0); assert P.prove(c, wm) == e1; c.close()
assert allocs(0) - a0 == cached, "key 1's idle tables on device 0 were evicted to make room on device 1"
assert L.zk_emul_current_device() == 0
c2.close(); k2.close()                                                              # frees tables that live on device 1 ...
assert L.zk_emul_current_device() == 0, "zk_pk_free left the caller on another device"
L.zk_emul_set_free_mem(0, C.c_size_t(1 << 20))                                      # ... and an eviction on device 0 does take device 0's idle set
k3 = P.ProvingKey.from_parts(**pk_o2.parts())
c3 = P.ProverContext(k3, r, device=0); assert P.prove(c3, wm) == e2; c3.close()
a0 = allocs(0)
c = P.ProverContext(k1, r, device=0); c.close()
assert allocs(0) - a0 > cached, "key 1's idle tables should have been evicted on their own device"
print("ok")
'''
    env = dict(os.environ, ZK_EMUL_DEVICES="2")
    out = subprocess.run([sys.executable, "-c", child, emul, ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.parametrize("seg", ["4", "256"])
def test_msm_row_column_reduction_shapes(zk, oracle, monkeypatch, seg):
    """the bucket reduction by row / column sums (k_msm_rowcol_sum, k_msm_weighted_sum): segment widths (threads per row), window sizes with
    square (c - 1 even) and 2 : 1 (c - 1 odd) bucket matrices, one and four lanes per thread, against the oracle"""
    monkeypatch.setenv("ZK_ROWCOL_SEG", seg)
    n = 220
    s = F.fr_to_mont([1] * 60 + rand_scalars(n - 60, 12, zeros_every=9))
    for g2 in (False, True):
        bases = oracle.batch_mul(F.fr_to_mont(rand_scalars(n, 23)), g2=g2)
        expect = oracle.msm(bases, s, g2=g2)
        for c, quad in ((2, "0"), (3, "1"), (8, "0"), (9, "1"), (9, "0")):
            monkeypatch.setenv("ZK_MSM_QUAD", quad)
            assert np.array_equal(zk.msm(bases, s, g2=g2, c=c), expect), (g2, c, quad)
