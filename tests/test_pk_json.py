"""The reference's second proving-key source: bellman / snarkjs style JSON -> nozk key -> `.raw`
(pk_bellman2ethsnarks, src/export.cpp:223-328).  The reference holds no such JSON file (parity unpinned by a
fixture); the check is three-way on a key made here: Python restatement (oracle/pyref.py) == library
(zk_pk_from_bellman_json) == the key the JSON was written from, and the `.raw` the converter writes is the oracle
codec's file byte for byte.  Host-only code: runs without a GPU."""
import json
import os
import numpy as np
import pytest
from conftest import ROOT
from ethsnarks_amd import r1cs as R, fields as F
import pyref

LIB = os.path.join(ROOT, "ethsnarks_amd", "libzkhip.so")
Q = F.FQ


def _jac1(p, rng):
    """affine (x, y) ints or None -> Jacobian decimal triple with a random z"""
    if p is None:
        return ["0", "1", "0"]
    z = rng.next() % Q or 1
    return [str(p[0] * z * z % Q), str(p[1] * pow(z, 3, Q) % Q), str(z)]


def _jac2(p, rng):
    if p is None:
        return [["0", "0"], ["1", "0"], ["0", "0"]]
    z = (rng.next() % Q, rng.next() % Q)
    z2 = pyref.f2_mul(z, z)
    x = pyref.f2_mul(p[0], z2); y = pyref.f2_mul(p[1], pyref.f2_mul(z2, z))
    return [[str(x[0]), str(x[1])], [str(y[0]), str(y[1])], [str(z[0]), str(z[1])]]


def _g1s(arr):
    v = F.fq_from_mont(np.ascontiguousarray(arr).reshape(-1, 4))
    return [None if (v[2 * i] == 0 and v[2 * i + 1] == 0) else (v[2 * i], v[2 * i + 1]) for i in range(len(v) // 2)]


def _g2s(arr):
    v = F.fq_from_mont(np.ascontiguousarray(arr).reshape(-1, 4))
    return [None if not any(v[4 * i:4 * i + 4]) else ((v[4 * i], v[4 * i + 1]), (v[4 * i + 2], v[4 * i + 3])) for i in range(len(v) // 4)]


def bellman_json_of(parts, seed=5):
    """bellman-form JSON text of a nozk key (one public input): dense A / B1 / B2 with zeros at absent indices, C = two
    placeholders + L; one extra trap entry: B2 non-zero where B1 is zero (must be dropped, export.cpp:294)."""
    rng = R.SplitMix64(seed)
    n = int(parts["a_domain"])
    a_idx = [int(i) for i in parts["a_idx"]]; b_idx = [int(i) for i in parts["b_idx"]]
    a_val = _g1s(parts["a_val"]); b_val = _g2s(parts["b_val"])
    A = [None] * n; B1 = [None] * n; B2 = [None] * n
    for i, p in zip(a_idx, a_val): A[i] = p
    some_g1 = _g1s(parts["alpha_g1"])[0]
    for i, p in zip(b_idx, b_val): B1[i] = some_g1; B2[i] = p
    trap = next((i for i in range(n) if B1[i] is None), None)
    if trap is not None:
        B2[trap] = _g2s(parts["beta_g2"])[0]
    d = {"A": [_jac1(p, rng) for p in A], "B1": [_jac1(p, rng) for p in B1], "B2": [_jac2(p, rng) for p in B2],
         "C": [_jac1(None, rng), _jac1(None, rng)] + [_jac1(p, rng) for p in _g1s(parts["L"])],
         "hExps": [_jac1(p, rng) for p in _g1s(parts["H"])],
         "vk_alfa_1": _jac1(_g1s(parts["alpha_g1"])[0], rng), "vk_beta_1": _jac1(_g1s(parts["beta_g1"])[0], rng),
         "vk_beta_2": _jac2(_g2s(parts["beta_g2"])[0], rng), "vk_delta_1": _jac1(_g1s(parts["delta_g1"])[0], rng),
         "vk_delta_2": _jac2(_g2s(parts["delta_g2"])[0], rng)}
    return json.dumps(d, indent=1), trap is not None


@pytest.fixture(scope="module")
def key(oracle):
    r, w = R.random_r1cs(24, 1, n_extra_vars=3, seed=9)        # unreferenced variables -> zero A / B entries, infinity L entries
    pk, vk = oracle.keygen(r, seed=3)
    return r, pk


def test_pyref_restatement_reads_the_key_back(key):
    r, pk = key
    P = pk.parts()
    text, trapped = bellman_json_of(P)
    assert trapped and len(P["a_idx"]) < P["a_domain"]          # the fixture exercises the zero-dropping rules
    got = pyref.pk_from_bellman_json(json.loads(text))
    assert got["A"] == ([int(i) for i in P["a_idx"]], _g1s(P["a_val"]))
    assert got["B"] == ([int(i) for i in P["b_idx"]], _g2s(P["b_val"]))
    assert got["H"] == _g1s(P["H"]) and got["L"] == _g1s(P["L"])
    assert got["alpha_g1"] == _g1s(P["alpha_g1"])[0] and got["delta_g2"] == _g2s(P["delta_g2"])[0]
    assert got["a_domain"] == P["a_domain"] == got["b_domain"]


@pytest.mark.skipif(not os.path.exists(LIB), reason="libzkhip.so not built (run __graft_entry__.build())")
def test_library_bellman_import_and_raw_conversion(key, tmp_path):
    from ethsnarks_amd import prover
    prover._lib = None; prover._lib_path_loaded = None
    prover.load_library()
    r, pk = key
    P = pk.parts()
    text, _ = bellman_json_of(P)
    jpath, rpath, opath = str(tmp_path / "pk.json"), str(tmp_path / "pk.raw"), str(tmp_path / "pk_oracle.raw")
    open(jpath, "w").write(text)
    got = prover.load_bellman_proving_key(jpath)
    G = got.parts()
    for k, v in P.items():
        if k == "b_domain":
            assert G[k] == P["a_domain"]                           # export.cpp:287: B's domain is |A|
        elif hasattr(v, "dtype"):
            assert np.array_equal(G[k], v), k
        else:
            assert G[k] == v, k
    assert prover.pk_bellman2ethsnarks(jpath, rpath) is True
    pk.write_raw(opath)
    if P["b_domain"] == P["a_domain"]:
        assert open(rpath, "rb").read() == open(opath, "rb").read()
    back = prover.load_proving_key(rpath)
    assert np.array_equal(back.parts()["b_val"], P["b_val"]) and np.array_equal(back.parts()["L"], P["L"])
    # error behaviour: missing file and malformed JSON are error codes, not aborts
    with pytest.raises(prover.ZkError):
        prover.load_bellman_proving_key(str(tmp_path / "missing.json"))
    open(jpath, "w").write(text.replace('"hExps"', '"hexps"'))
    with pytest.raises(prover.ZkError):
        prover.load_bellman_proving_key(jpath)
    open(jpath, "w").write(text.replace('"1"', '"1x"', 1))
    with pytest.raises(prover.ZkError):
        prover.load_bellman_proving_key(jpath)
    prover._lib = None; prover._lib_path_loaded = None


def _full_key_stream(P):
    """test-side writer of the FULL (zero-knowledge) proving key stream of r1cs_gg_ppzksnark_zok.tcc:53-66 from a nozk
    key: A_query gets its zero entries back, every B_query value becomes a knowledge commitment (g = the G2 value, h = a
    G1 point made up here: beta_g1).  Point = '0' + raw Montgomery coordinates, infinity = '1' + (0, one)."""
    one = np.ascontiguousarray(F.fq_to_mont([1])).reshape(-1)

    def pt(raw, nlimb):
        raw = np.ascontiguousarray(raw, dtype=np.uint64).reshape(-1)
        if not raw.any():
            img = np.zeros(nlimb, dtype=np.uint64); img[nlimb // 2:nlimb // 2 + 4] = one
            return b"1" + img.tobytes()
        return b"0" + raw.tobytes()
    out = [pt(P["alpha_g1"], 8), pt(P["beta_g1"], 8), pt(P["beta_g2"], 16), pt(P["delta_g1"], 8), pt(P["delta_g2"], 16)]
    n = int(P["a_domain"])
    A = np.zeros((n, 8), dtype=np.uint64)
    A[np.asarray(P["a_idx"], dtype=np.int64)] = P["a_val"]
    out.append(b"%d\n" % n + b"".join(pt(A[i], 8) for i in range(n)))
    out.append(b"%d\n%d\n" % (int(P["b_domain"]), len(P["b_idx"])) + b"".join(b"%d\n" % int(i) for i in P["b_idx"]) + b"%d\n" % len(P["b_idx"]))
    out.append(b"".join(pt(P["b_val"][i], 16) + pt(P["beta_g1"], 8) for i in range(len(P["b_idx"]))))
    for q in ("H", "L"):
        out.append(b"%d\n" % len(P[q]) + b"".join(pt(P[q][i], 8) for i in range(len(P[q]))))
    return b"".join(out)


@pytest.mark.skipif(not os.path.exists(LIB), reason="libzkhip.so not built (run __graft_entry__.build())")
def test_mcl_codec_and_the_alt2mcl_mcl2nozk_converters(key, tmp_path):
    """ZK_CODEC_MCL_BN128 (the reference's default curve build; element layout INFERRED, parity unpinned: the reference holds
    no key file) and the offline converters pk_alt2mcl / pk_mcl2nozk (src/export.cpp:330-408): a full key goes alt -> mcl
    through the decimal strings of every coordinate, mcl -> nozk drops A's zeros and B's G1 halves, and the nozk key read
    back with the MCL codec is the key the chain started from."""
    from ethsnarks_amd import prover
    prover._lib = None; prover._lib_path_loaded = None
    prover.load_library()
    r, pk = key
    P = pk.parts()
    alt, mcl, nozk, ref = (str(tmp_path / n) for n in ("full_alt.raw", "full_mcl.raw", "nozk_mcl.raw", "nozk_oracle.raw"))
    open(alt, "wb").write(_full_key_stream(P))
    assert prover.pk_alt2mcl(alt, mcl) is True
    assert open(mcl, "rb").read() == open(alt, "rb").read()      # the decimal path lands on the inferred layout's bytes
    assert prover.pk_mcl2nozk(mcl, nozk) is True
    pk.write_raw(ref)
    assert open(nozk, "rb").read() == open(ref, "rb").read()
    back = prover.load_proving_key(nozk, codec=prover.CODEC_MCL_BN128)
    G = back.parts()
    for k, v in P.items():
        assert np.array_equal(G[k], v) if hasattr(v, "dtype") else G[k] == v, k
    rt = str(tmp_path / "rt.raw")
    back.save_raw(rt, codec=prover.CODEC_MCL_BN128)
    assert open(rt, "rb").read() == open(ref, "rb").read()
    with pytest.raises(prover.ZkError) as e:
        back.save_raw(rt, codec=2)
    assert e.value.code == 1
    open(alt, "wb").write(_full_key_stream(P)[:-7])              # truncated stream: an error code, not an abort
    with pytest.raises(prover.ZkError) as e:
        prover.pk_alt2mcl(alt, mcl)
    assert e.value.code == 3
    prover._lib = None; prover._lib_path_loaded = None
