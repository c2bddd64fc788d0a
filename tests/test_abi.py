"""The C-ABI library loads and exports every symbol include/zkhip.h declares; without a GPU every
compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import numpy as np
import pytest
from conftest import ROOT, gpu_available

LIB = os.path.join(ROOT, "ethsnarks_amd", "libzkhip.so")


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", hdr)))          # + ethsnarks_verify, checked below


def test_header_symbols_are_bound_in_python():
    from ethsnarks_amd import prover
    assert sorted(prover.EXPORTS) == declared_symbols()


@pytest.mark.skipif(not os.path.exists(LIB), reason="libzkhip.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    L = C.CDLL(LIB)
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert hasattr(L, "ethsnarks_verify")              # the reference's own verify symbol (src/verify_dll.cpp:3-10)
    L.zk_version.restype = C.c_char_p
    assert b"gfx950" in L.zk_version()


@pytest.mark.skipif(not os.path.exists(LIB) or gpu_available(), reason="needs the built library and NO gpu")
def test_compute_fails_loudly_without_device():
    from ethsnarks_amd import prover
    prover._lib = None; prover._lib_path_loaded = None
    prover.load_library()
    a = np.zeros((4, 4), dtype=np.uint64)
    with pytest.raises(prover.ZkError) as e:
        prover.field_mul(a, a)
    assert e.value.code in (4, 8)
    with pytest.raises(prover.ZkError):
        prover.ntt(a, 2)
    prover._lib = None; prover._lib_path_loaded = None


def test_missing_library_is_an_import_error(tmp_path):
    from ethsnarks_amd import prover
    prover._lib = None; prover._lib_path_loaded = None
    with pytest.raises(ImportError):
        prover.load_library(str(tmp_path / "libzkhip.so"))


@pytest.mark.skipif(not os.path.exists(LIB), reason="libzkhip.so not built (run __graft_entry__.build())")
def test_plain_c_client_links_and_verifies(tmp_path):
    """include/zkhip.h is a C header: a C99 program links libzkhip.so and runs the host-side verifier on the
    reference's static triple (no GPU involved)."""
    import json, subprocess
    exe = str(tmp_path / "c_abi_client")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_client.c"), "-o", exe, LIB,
                           "-Wl,-rpath," + os.path.dirname(LIB), "-Wl,-rpath,/opt/rocm/lib"])
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_static_triple.json")))
    vk, pf, bad = tmp_path / "vk.json", tmp_path / "proof.json", tmp_path / "bad.json"
    vk.write_text(json.dumps(d["vk"])); pf.write_text(json.dumps(d["proof"]))
    bad.write_text(json.dumps(dict(d["proof"], input=[d["proof"]["input"][0], "0x8"])))
    out = subprocess.run([exe, str(vk), str(pf)], capture_output=True, text=True)
    assert out.returncode == 0 and "accepted" in out.stdout and "= 1048576" in out.stdout, out.stdout + out.stderr
    out = subprocess.run([exe, str(vk), str(bad)], capture_output=True, text=True)
    assert out.returncode == 1 and "rejected" in out.stdout


@pytest.mark.skipif(not os.path.exists(LIB), reason="libzkhip.so not built (run __graft_entry__.build())")
def test_json_readers_and_writers_on_the_reference_static_triple():
    """host-only entry points, no GPU needed: the reference's static proof / key text (test/test_verify.py:10-12) goes
    through zk_proof_from_json / zk_vk_from_json and comes back from zk_proof_to_json / zk_vk_to_json token for token"""
    import json
    from ethsnarks_amd import prover
    prover._lib = None; prover._lib_path_loaded = None
    prover.load_library()
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_static_triple.json")))
    flat = lambda x: [x] if isinstance(x, str) else [t for y in x for t in flat(y)]
    proof, inputs = prover.proof_from_json(json.dumps(d["proof"]))
    text = prover.proof_to_json(proof, inputs, canonical=True)
    assert json.loads(text) == d["proof"]
    assert re.findall(r'"(0x[0-9a-f]+)"', text) == flat([d["proof"][k] for k in ("A", "B", "C", "input")])
    vk = prover.vk_from_json(json.dumps(d["vk"]))
    assert json.loads(vk.to_json()) == d["vk"]
    assert re.findall(r'"(0x[0-9a-f]+)"', vk.to_json()) == flat([d["vk"][k] for k in ("alpha", "beta", "gamma", "delta", "gammaABC")])
    # decimal strings are accepted as well (parse_bigint, src/import.hpp:15-33); coordinates >= q are not
    dec = json.loads(json.dumps(d["proof"]))
    dec["A"] = [str(int(t, 16)) for t in dec["A"]]
    p2, _ = prover.proof_from_json(json.dumps(dec))
    assert bytes(p2) == bytes(proof)
    bad = dict(d["proof"], A=["0x" + "f" * 64, d["proof"]["A"][1]])
    with pytest.raises(prover.ZkError) as e:
        prover.proof_from_json(json.dumps(bad))
    assert e.value.code == 3
    prover._lib = None; prover._lib_path_loaded = None


@pytest.mark.skipif(not os.path.exists(LIB), reason="libzkhip.so not built (run __graft_entry__.build())")
def test_unknown_raw_codec_is_rejected(tmp_path):
    from ethsnarks_amd import prover
    prover._lib = None; prover._lib_path_loaded = None
    prover.load_library()
    with pytest.raises(prover.ZkError) as e:
        prover.load_proving_key(str(tmp_path / "pk.raw"), codec=77)
    assert e.value.code == 1
    prover._lib = None; prover._lib_path_loaded = None


@pytest.mark.skipif(not os.path.exists(LIB), reason="libzkhip.so not built (run __graft_entry__.build())")
def test_abi_version_and_sized_config():
    """ZK_ABI_VERSION of the header == zk_abi_version() of the library == the binding's; zk_ctx_create_sized takes the caller's
    sizeof(zk_config) (argument checks run before any device work, so this needs no GPU: a null key is ZK_ERR_ARG either way)"""
    import re
    from ethsnarks_amd import prover
    L = C.CDLL(LIB)
    L.zk_abi_version.restype = C.c_uint32
    header = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    assert int(re.search(r"#define ZK_ABI_VERSION (\d+)", header).group(1)) == L.zk_abi_version() == prover.ABI_VERSION
    assert C.sizeof(prover.ZkConfig) == 24                     # the layout ABI version 3 names
    cfg = prover.ZkConfig(0, 0, 0, 1, 1, 0)
    out = C.c_void_p()
    assert L.zk_ctx_create_sized(None, None, None, None, 0, 0, 0, C.byref(cfg), C.c_size_t(16), C.byref(out)) == 1   # ZK_ERR_ARG (null key), not a crash
