"""Host-side binding of libzkhip.so (C ABI: include/zkhip.h) for tests/ and bench.py.

The names mirror the reference's proving API (src/stubs.hpp:18-21, src/stubs.cpp:36-75):

    load_proving_key(path)            -> ProvingKey        ethsnarks::load_proving_key
    ProverContext(pk, r1cs, config)   -> context           ProverContext<ppT> + get_domain
    prove(ctx, witness)               -> proof JSON str    ethsnarks::prove
    stub_prove_from_pb(r1cs, w, path) -> proof JSON str    upstream wrapper used by src/pinocchio/main.cpp:41

The reference itself is C++; the C++ adapter with the same signatures lives in
include/ethsnarks_hip/stubs.hpp.  This module adds nothing to the data path: it passes numpy buffers
to the C ABI.  It fails loudly when the HIP library is missing -- there is no CPU fallback.
"""
import ctypes as C
import os
import numpy as np

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libzkhip.so")


class ZkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("zkhip error %d: %s" % (code, msg))
        self.code = code


class ZkCSR(C.Structure):
    _fields_ = [("n_rows", C.c_uint32), ("row_ptr", _u32p), ("col", _u32p), ("coeff", _u64p)]


class ZkConfig(C.Structure):
    """libsnark::Config (src/prover_config.hpp:8-35) reduced to what a GPU prover consumes."""
    _fields_ = [("multi_exp_c", C.c_uint32), ("device", C.c_uint32), ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32), ("max_batch", C.c_uint32), ("schedule", C.c_uint32)]


class ZkProof(C.Structure):
    _fields_ = [(n, C.c_uint64 * 4) for n in
                ("a_x", "a_y", "b_x_c0", "b_x_c1", "b_y_c0", "b_y_c1", "c_x", "c_y")] + \
               [("a_inf", C.c_uint32), ("b_inf", C.c_uint32), ("c_inf", C.c_uint32), ("_pad", C.c_uint32)]


class ZkPartials(C.Structure):
    _fields_ = [("At", C.c_uint64 * 16), ("Bt", C.c_uint64 * 32), ("Ht", C.c_uint64 * 16), ("Lt", C.c_uint64 * 16)]


class ZkTimings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("h2d_witness", "compute_h", "a_query", "b_query", "h_query", "l_query", "gpu_total", "host_finish",
                                             "acc_a", "acc_b", "acc_h", "acc_l")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


ABI_VERSION = 3          # include/zkhip.h ZK_ABI_VERSION

EXPORTS = [
    "zk_version", "zk_abi_version", "zk_strerror", "zk_last_error", "zk_device_count",
    "zk_pk_load_raw", "zk_pk_save_raw", "zk_pk_from_bellman_json", "zk_pk_bellman2ethsnarks", "zk_pk_alt2mcl", "zk_pk_mcl2nozk", "zk_pk_from_parts", "zk_pk_sizes", "zk_pk_part", "zk_pk_free",
    "zk_keygen", "zk_vk_to_json", "zk_vk_from_json", "zk_proof_from_json", "zk_vk_free",
    "zk_domain_size", "zk_ctx_create", "zk_ctx_create_sized", "zk_ctx_destroy",
    "zk_prove", "zk_prove_timed", "zk_prove_partial", "zk_prove_partial_timed", "zk_prove_combine", "zk_prove_submit", "zk_prove_collect", "zk_proof_to_json",
    "zk_prove_batch", "zk_prove_batch_submit", "zk_prove_batch_submit_resident", "zk_prove_batch_collect",
    "zk_wplan_create", "zk_wplan_create_hinted", "zk_wplan_solve", "zk_wplan_free", "zk_dev_alloc", "zk_dev_free", "zk_dev_upload", "zk_dev_download",
    "zk_chain_submit", "zk_chain_device", "zk_h_from_chains_submit", "zk_h_device", "zk_chain_wait", "zk_prove_submit_with_h", "zk_prove_submit_defer_h", "zk_prove_submit_h", "zk_prove_abort",
    "zk_prove_submit_pinned", "zk_prove_batch_submit_pinned", "zk_host_alloc", "zk_host_free", "zk_host_register", "zk_host_unregister",
    "zk_prove_submit_resident", "zk_prove_stage", "zk_prove_stage_pinned", "zk_prove_submit_staged", "zk_ctx_info", "zk_ctx_table_info", "zk_ctx_partials_device", "zk_prove_collect_device", "zk_prove_combine_device", "zk_launch_count", "zk_profile_begin", "zk_profile_end", "zk_device_info", "zk_device_pci_bus_id",
    "zk_verify",
    "zk_ntt", "zk_witness_map", "zk_msm_g1", "zk_msm_g2", "zk_field_mul", "zk_fr_convert",
]

_lib = None
_lib_path_loaded = None
_legacy_abi = False


def load_library(path=None):
    """Load libzkhip.so (or an explicitly named build of the same ABI).  Raises if it is missing."""
    global _lib, _lib_path_loaded
    path = path or LIB_PATH
    if _lib is not None and _lib_path_loaded == path:
        return _lib
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    L = C.CDLL(path)
    for n in ("zk_version", "zk_strerror", "zk_last_error"):
        getattr(L, n).restype = C.c_char_p
    L.zk_pk_part.restype = C.c_void_p
    L.zk_domain_size.restype = C.c_uint32
    L.zk_launch_count.restype = C.c_uint64
    L.zk_ctx_partials_device.restype = C.c_void_p
    L.zk_chain_device.restype = C.c_void_p
    L.zk_h_device.restype = C.c_void_p
    global _legacy_abi
    try:
        L.zk_abi_version.restype = C.c_uint32
        have = L.zk_abi_version()
    except AttributeError:
        have = 2                                           # rounds 1-2: no zk_abi_version; zk_config had the same six members in round 2
    _legacy_abi = have != ABI_VERSION
    if _legacy_abi and not (have == 2 and os.environ.get("ZK_LIB_ALLOW_OLD_ABI") == "1"):   # (dev aid: same-box A/B against a round-2 build)
        raise ImportError("%s has ABI version %d, this binding was written for %d" % (path, have, ABI_VERSION))   # struct layouts are those of include/zkhip.h
    _lib, _lib_path_loaded = L, path
    return L


def _check(rc):
    if rc != 0:
        L = load_library(_lib_path_loaded)
        raise ZkError(rc, (L.zk_last_error() or L.zk_strerror(rc)).decode())


def _p64(a):
    return a.ctypes.data_as(_u64p)


def _p32(a):
    return a.ctypes.data_as(_u32p)


def _c64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a if shape is None else a.reshape(shape)


def device_count():
    n = C.c_int(0)
    rc = load_library(_lib_path_loaded).zk_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class ProvingKey:
    """ethsnarks::ProvingKeyT = r1cs_gg_ppzksnark_zok_proving_key_nozk (hpp:171-274): host object."""

    def __init__(self, handle):
        self._h = handle
        s = (C.c_uint32 * 6)()
        _check(_lib.zk_pk_sizes(handle, s))
        self.a_domain, self.nA, self.b_domain, self.nB, self.nH, self.nL = list(s)

    @staticmethod
    def from_parts(alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, a_domain, a_idx, a_val, b_domain, b_idx, b_val, H, L):
        lib = load_library(_lib_path_loaded)
        a_idx = np.ascontiguousarray(a_idx, dtype=np.uint32); b_idx = np.ascontiguousarray(b_idx, dtype=np.uint32)
        a_val = _c64(a_val); b_val = _c64(b_val); H = _c64(H); L = _c64(L)
        h = C.c_void_p()
        _check(lib.zk_pk_from_parts(_p64(_c64(alpha_g1)), _p64(_c64(beta_g1)), _p64(_c64(beta_g2)), _p64(_c64(delta_g1)), _p64(_c64(delta_g2)),
                                    C.c_uint32(a_domain), C.c_uint32(len(a_idx)), _p32(a_idx), _p64(a_val),
                                    C.c_uint32(b_domain), C.c_uint32(len(b_idx)), _p32(b_idx), _p64(b_val),
                                    C.c_uint32(H.size // 8), _p64(H), C.c_uint32(L.size // 8), _p64(L), C.byref(h)))
        return ProvingKey(h)

    def part(self, which, shape, dtype=np.uint64):
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype=dtype)
        p = _lib.zk_pk_part(self._h, which)
        ct = (C.c_uint64 if dtype == np.uint64 else C.c_uint32) * n
        return np.frombuffer(ct.from_address(p), dtype=dtype).reshape(shape).copy()

    def parts(self):
        return dict(alpha_g1=self.part(0, (8,)), beta_g1=self.part(1, (8,)), beta_g2=self.part(2, (16,)),
                    delta_g1=self.part(3, (8,)), delta_g2=self.part(4, (16,)),
                    a_domain=self.a_domain, a_idx=self.part(5, (self.nA,), np.uint32), a_val=self.part(6, (self.nA, 8)),
                    b_domain=self.b_domain, b_idx=self.part(7, (self.nB,), np.uint32), b_val=self.part(8, (self.nB, 16)),
                    H=self.part(9, (self.nH, 8)), L=self.part(10, (self.nL, 8)))

    def save_raw(self, path, codec=0):
        """writeToFile<ProvingKeyT> (src/utils.hpp:166-173)."""
        _check(_lib.zk_pk_save_raw(self._h, os.fsencode(path), codec))

    def close(self):
        if self._h is not None and _lib is not None:
            _lib.zk_pk_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_proving_key(pk_file, codec=0):
    """ethsnarks::load_proving_key (src/stubs.cpp:36-39).  A missing file is an error here
    (the reference asserts, src/utils.hpp:180)."""
    lib = load_library(_lib_path_loaded)
    h = C.c_void_p()
    _check(lib.zk_pk_load_raw(os.fsencode(pk_file), codec, C.byref(h)))
    return ProvingKey(h)


def load_bellman_proving_key(json_file):
    """proving key from a bellman / snarkjs style JSON, mapped as pk_bellman2ethsnarks does (src/export.cpp:267-321)."""
    lib = load_library(_lib_path_loaded)
    h = C.c_void_p()
    _check(lib.zk_pk_from_bellman_json(os.fsencode(json_file), C.byref(h)))
    return ProvingKey(h)


def pk_bellman2ethsnarks(bellman_pk_file, pk_file):
    """pk_bellman2ethsnarks (src/export.cpp:267-328): JSON in, nozk `.raw` out; True on success like the reference."""
    _check(load_library(_lib_path_loaded).zk_pk_bellman2ethsnarks(os.fsencode(bellman_pk_file), os.fsencode(pk_file)))
    return True


CODEC_ALT_BN128, CODEC_MCL_BN128 = 0, 1     # include/zkhip.h; the MCL element layout is inferred (parity unpinned)


def pk_alt2mcl(alt_pk_file, mcl_pk_file):
    """pk_alt2mcl (src/export.cpp:352-397) over the full proving-key stream."""
    _check(load_library(_lib_path_loaded).zk_pk_alt2mcl(os.fsencode(alt_pk_file), os.fsencode(mcl_pk_file)))
    return True


def pk_mcl2nozk(mcl_pk_file, nozk_pk_file):
    """pk_mcl2nozk (src/export.cpp:399-408)."""
    _check(load_library(_lib_path_loaded).zk_pk_mcl2nozk(os.fsencode(mcl_pk_file), os.fsencode(nozk_pk_file)))
    return True


class VerificationKey:
    """r1cs_gg_ppzksnark_zok_verification_key (hpp:296-350); only what keygen hands out."""

    def __init__(self, handle):
        self._h = handle

    def to_json(self):
        """vk2json (src/export.cpp:124-145)."""
        ln = C.c_size_t(0)
        _lib.zk_vk_to_json(self._h, None, C.c_size_t(0), C.byref(ln))
        buf = C.create_string_buffer(ln.value + 1)
        _check(_lib.zk_vk_to_json(self._h, buf, C.c_size_t(ln.value + 1), C.byref(ln)))
        return buf.raw[:ln.value].decode()

    def close(self):
        if self._h is not None and _lib is not None:
            _lib.zk_vk_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vk_from_json(text):
    """vk_from_json (src/import.cpp:195-223): hex / decimal strings, Fq2 as [c1, c0]."""
    h = C.c_void_p()
    _check(load_library(_lib_path_loaded).zk_vk_from_json(text.encode(), C.byref(h)))
    return VerificationKey(h)


def proof_from_json(text):
    """proof_from_json (src/import.cpp:161-192): (ZkProof, canonical inputs[n, 4])."""
    lib = load_library(_lib_path_loaded)
    proof, n = ZkProof(), C.c_uint32(0)
    rc = lib.zk_proof_from_json(text.encode(), C.byref(proof), None, C.c_uint32(0), C.byref(n))
    if rc != 0 and n.value == 0:
        _check(rc)
    inputs = np.zeros((n.value, 4), dtype=np.uint64)
    _check(lib.zk_proof_from_json(text.encode(), C.byref(proof), _p64(inputs) if n.value else None, C.c_uint32(n.value), C.byref(n)))
    return proof, inputs


def _csr_structs(r1cs, keep):
    def csr(m):
        rp = np.ascontiguousarray(m.row_ptr, dtype=np.uint32)
        co = np.ascontiguousarray(m.col, dtype=np.uint32)
        cf = _c64(m.coeff)
        keep.extend([rp, co, cf])
        return ZkCSR(m.n_rows, _p32(rp), _p32(co), _p64(cf))
    return csr(r1cs.A), csr(r1cs.B), csr(r1cs.C)


def keygen(r1cs, toxic=None, seed=None, device=0):
    """r1cs_gg_ppzksnark_zok_generator + nozk conversion (tcc:277-449, hpp:209-233) on the GPU.
    toxic = (t, alpha, beta, gamma, delta) ints; or seed -> 5 SplitMix64 draws; default os.urandom."""
    lib = load_library(_lib_path_loaded)
    from .fields import FR, ints_to_limbs
    if toxic is None:
        if seed is not None:
            from .r1cs import SplitMix64
            rng = SplitMix64(seed)
            toxic = [rng.fr() for _ in range(5)]
        else:
            toxic = [int.from_bytes(os.urandom(40), "little") % FR for _ in range(5)]
    tox = np.ascontiguousarray(ints_to_limbs([t % FR for t in toxic])).reshape(-1)
    keep = []
    a, b, c = _csr_structs(r1cs, keep)
    pk, vk = C.c_void_p(), C.c_void_p()
    _check(lib.zk_keygen(C.byref(a), C.byref(b), C.byref(c), C.c_uint32(r1cs.nC), C.c_uint32(r1cs.nIn), C.c_uint32(r1cs.V),
                         _p64(tox), device, C.byref(pk), C.byref(vk)))
    return ProvingKey(pk), VerificationKey(vk)


def stub_genkeys_from_pb(r1cs, pk_file, vk_file, **kw):
    """ethsnarks::stub_genkeys_from_pb (src/stubs.cpp:77-87): vk JSON + nozk .raw proving key."""
    pk, vk = keygen(r1cs, **kw)
    with open(vk_file, "wb") as f:                     # vk2json_file, src/export.cpp:148-155
        f.write(vk.to_json().encode())
    pk.save_raw(pk_file)
    return 0


def get_domain_size(r1cs):
    """size of the domain ethsnarks::get_domain builds (src/stubs.cpp:61-75)."""
    return int(load_library(_lib_path_loaded).zk_domain_size(C.c_uint32(r1cs.nC), C.c_uint32(r1cs.nIn)))


class ProverContext:
    """ProverContext<ppT> (hpp:279-291) + get_domain: bases and CSR resident in HBM, scratch owned."""

    def __init__(self, pk, r1cs, multi_exp_c=0, device=0, shard_rank=0, shard_count=1, max_batch=1, one_stream=False, latency=False):
        """one_stream: zk_config.schedule = ZK_SCHED_ONE_STREAM (many small proofs in many contexts); latency: ZK_SCHED_LATENCY (a context
        that proves one synchronous proof at a time, what ethsnarks::prove creates)"""
        lib = load_library(_lib_path_loaded)
        self.r1cs = r1cs
        self.max_batch = max_batch
        self._keep = []
        a, b, c = _csr_structs(r1cs, self._keep)
        cfg = ZkConfig(multi_exp_c, device, shard_rank, shard_count, max_batch, 1 if one_stream else 2 if latency else 0)
        h = C.c_void_p()
        if _legacy_abi:
            _check(lib.zk_ctx_create(pk._h, C.byref(a), C.byref(b), C.byref(c), C.c_uint32(r1cs.nC), C.c_uint32(r1cs.nIn), C.c_uint32(r1cs.V), C.byref(cfg), C.byref(h)))
        else:
            _check(lib.zk_ctx_create_sized(pk._h, C.byref(a), C.byref(b), C.byref(c), C.c_uint32(r1cs.nC), C.c_uint32(r1cs.nIn),
                                           C.c_uint32(r1cs.V), C.byref(cfg), C.c_size_t(C.sizeof(cfg)), C.byref(h)))
        self._h = h
        self._keep = []     # the context copied everything it needs
        self.shard_count = shard_count

    def _w(self, witness):
        w = _c64(witness)
        if w.size != 4 * (self.r1cs.V + 1):
            raise ValueError("witness must have V + 1 = %d elements" % (self.r1cs.V + 1))
        return w

    def prove_struct(self, witness, canonical=False, timings=False):
        w = self._w(witness)
        proof = ZkProof()
        if timings:
            t = ZkTimings()
            _check(_lib.zk_prove_timed(self._h, _p64(w), int(canonical), C.byref(proof), C.byref(t)))
            return proof, t.as_dict()
        _check(_lib.zk_prove(self._h, _p64(w), int(canonical), C.byref(proof)))
        return proof

    def prove_partial(self, witness, canonical=False, timings=False):
        w = self._w(witness)
        part = ZkPartials()
        if timings:
            t = ZkTimings()
            _check(_lib.zk_prove_partial_timed(self._h, _p64(w), int(canonical), C.byref(part), C.byref(t)))
            return np.frombuffer(bytes(part), dtype=np.uint64).copy(), t.as_dict()
        _check(_lib.zk_prove_partial(self._h, _p64(w), int(canonical), C.byref(part)))
        return np.frombuffer(bytes(part), dtype=np.uint64).copy()      # 80 u64 = 640 bytes

    def submit(self, witness, canonical=False):
        """enqueue a proof and return (zk_prove_submit); the witness is copied to pinned memory first"""
        w = self._w(witness)
        _check(_lib.zk_prove_submit(self._h, _p64(w), int(canonical)))

    def _wk(self, witnesses):
        w = _c64(witnesses)
        per = 4 * (self.r1cs.V + 1)
        if w.size % per or w.size == 0:
            raise ValueError("witnesses must be k x (V + 1) elements")
        return w, w.size // per

    def prove_batch_structs(self, witnesses, canonical=False):
        """zk_prove_batch: k witnesses of this circuit through one launch sequence -> list of k ZkProof"""
        w, k = self._wk(witnesses)
        proofs = (ZkProof * k)()
        _check(_lib.zk_prove_batch(self._h, _p64(w), C.c_uint32(k), int(canonical), proofs))
        return list(proofs)

    def submit_batch(self, witnesses, canonical=False, device_ptr=None, k=None):
        if device_ptr is not None:
            _check(_lib.zk_prove_batch_submit_resident(self._h, C.c_void_p(device_ptr), C.c_uint32(k), int(canonical)))
            return k
        w, k = self._wk(witnesses)
        _check(_lib.zk_prove_batch_submit(self._h, _p64(w), C.c_uint32(k), int(canonical)))
        return k

    def collect_batch(self, k):
        """(partials[k, 80 u64], timings)"""
        parts, t = (ZkPartials * k)(), ZkTimings()
        _check(_lib.zk_prove_batch_collect(self._h, parts, C.c_uint32(k), C.byref(t)))
        return np.frombuffer(bytes(parts), dtype=np.uint64).reshape(k, 80).copy(), t.as_dict()

    def submit_resident(self, device_ptr, canonical=False):
        """zk_prove_submit_resident: the witness already lives in this device's memory (device_ptr = integer address of
        (V + 1) x 32 bytes); the caller keeps that buffer untouched until collect()"""
        _check(_lib.zk_prove_submit_resident(self._h, C.c_void_p(device_ptr), int(canonical)))

    def submit_pinned(self, pinned, canonical=False, k=1):
        """zk_prove_submit_pinned / zk_prove_batch_submit_pinned: `pinned` is a PinnedBuffer (or a pinned numpy view) holding k
        witnesses; the H2D copy reads it in place, so it must stay untouched until the proof is collected"""
        lib = load_library(_lib_path_loaded)
        ptr = C.c_void_p(pinned.ptr if isinstance(pinned, PinnedBuffer) else pinned.ctypes.data)
        if k == 1:
            _check(lib.zk_prove_submit_pinned(self._h, ptr, int(canonical)))
        else:
            _check(lib.zk_prove_batch_submit_pinned(self._h, ptr, C.c_uint32(k), int(canonical)))
        return k

    def stage_pinned(self, pinned, canonical=False, k=1):
        """zk_prove_stage_pinned: the NEXT proof's k witnesses go to the device from a PinnedBuffer, in place, on the copy stream"""
        ptr = C.c_void_p(pinned.ptr if isinstance(pinned, PinnedBuffer) else pinned.ctypes.data)
        _check(load_library(_lib_path_loaded).zk_prove_stage_pinned(self._h, ptr, C.c_uint32(k), int(canonical)))

    def stage(self, witnesses, canonical=False):
        """zk_prove_stage: copy the NEXT witness (or k of them, shape (k, V + 1, 4)) to the device while a proof may still be
        in flight on this context; returns k"""
        w = np.ascontiguousarray(witnesses, dtype=np.uint64)
        k = 1 if w.ndim == 2 else int(w.shape[0])
        _check(_lib.zk_prove_stage(self._h, _p64(w), C.c_uint32(k), int(canonical)))
        return k

    def submit_staged(self):
        """zk_prove_submit_staged: start the staged proof (no upload on its critical path); collect() / collect_batch(k) as usual"""
        _check(_lib.zk_prove_submit_staged(self._h))

    def info(self):
        """window bits / windows / buckets per query, shared-sort flags, domain size (zk_ctx_info)"""
        a = (C.c_uint32 * 16)()
        _check(_lib.zk_ctx_info(self._h, a))
        v = list(a)
        d = {q: {"c": v[3 * i], "W": v[3 * i + 1], "buckets": v[3 * i + 2]} for i, q in enumerate("ABHL")}
        d.update(share_A=bool(v[12]), share_B=bool(v[13]), share_L=bool(v[14]), m=v[15])
        t = (C.c_uint64 * 4)()
        _check(_lib.zk_ctx_table_info(self._h, t))                      # planes > 1: memory-frugal tables (every planes-th window tabulated)
        d.update(table_bytes=int(t[0]), full_table_bytes=int(t[1]), planes=int(t[2]), table_rows_B=int(t[3]))
        return d

    def collect(self):
        """wait for the submitted proof: (partials[80 u64], timings dict)"""
        part, t = ZkPartials(), ZkTimings()
        _check(_lib.zk_prove_collect(self._h, C.byref(part), C.byref(t)))
        return np.frombuffer(bytes(part), dtype=np.uint64).copy(), t.as_dict()

    # SURVEY 8(e) option 2 building blocks (ethsnarks_amd/sharded.py drives them)
    def chain_submit(self, witness, which, canonical=False):
        """witness None: the chain of the proof submitted with submit_defer_h (its witness is already on the device)"""
        _check(_lib.zk_chain_submit(self._h, _p64(self._w(witness)) if witness is not None else None, int(canonical), int(which)))

    def submit_defer_h(self, witness, canonical=False):
        _check(_lib.zk_prove_submit_defer_h(self._h, _p64(self._w(witness)), int(canonical)))

    def submit_h(self, h_device_ptr):
        _check(_lib.zk_prove_submit_h(self._h, C.c_void_p(h_device_ptr)))

    def abort(self):
        _check(_lib.zk_prove_abort(self._h))

    def chain_device_ptr(self, which):
        return int(_lib.zk_chain_device(self._h, int(which)))

    def h_from_chains_submit(self, pa, pb, pc):
        _check(_lib.zk_h_from_chains_submit(self._h, C.c_void_p(pa), C.c_void_p(pb), C.c_void_p(pc)))

    def h_device_ptr(self):
        return int(_lib.zk_h_device(self._h))

    def chain_wait(self, check_degree=False):
        _check(_lib.zk_chain_wait(self._h, int(check_degree)))

    def submit_with_h(self, witness, h_device_ptr, canonical=False):
        _check(_lib.zk_prove_submit_with_h(self._h, _p64(self._w(witness)), int(canonical), C.c_void_p(h_device_ptr)))

    def partials_device_ptr(self):
        """address of the context's 640-byte device copy of the partial sums (valid after collect_device)"""
        return int(_lib.zk_ctx_partials_device(self._h))

    def collect_device(self):
        """wait for the submitted proof, leave the partial sums on the device: timings dict"""
        t = ZkTimings()
        _check(_lib.zk_prove_collect_device(self._h, C.byref(t)))
        return t.as_dict()

    def prove_combine_device(self, device_ptr, count):
        """fold `count` gathered 640-byte records that live in device memory (zk_prove_combine_device)"""
        proof = ZkProof()
        _check(_lib.zk_prove_combine_device(self._h, C.c_void_p(device_ptr), C.c_uint32(count), C.byref(proof)))
        return proof

    def prove_combine(self, partials):
        arr = _c64(partials).reshape(-1, 80)
        proof = ZkProof()
        _check(_lib.zk_prove_combine(self._h, C.cast(_p64(arr), C.POINTER(ZkPartials)), C.c_uint32(arr.shape[0]), C.byref(proof)))
        return proof

    def witness_map(self, witness, canonical=False):
        w = self._w(witness)
        m = self.r1cs.domain_size
        h = np.zeros((m + 1, 4), dtype=np.uint64)
        _check(_lib.zk_witness_map(self._h, _p64(w), int(canonical), _p64(h)))
        return h

    def close(self):
        if getattr(self, "_h", None) is not None and _lib is not None:
            _lib.zk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def proof_to_json(proof, inputs, canonical=False):
    """ethsnarks::proof_to_json (src/export.cpp:99-121): exact text."""
    inputs = _c64(inputs).reshape(-1, 4)
    n = inputs.shape[0]
    cap = 2048 + 80 * n
    buf = C.create_string_buffer(cap)
    ln = C.c_size_t(0)
    _check(_lib.zk_proof_to_json(C.byref(proof), _p64(inputs) if n else None, C.c_uint32(n), int(canonical), buf, C.c_size_t(cap), C.byref(ln)))
    return buf.raw[:ln.value].decode()


def prove_batch(ctx, witnesses, canonical=False):
    """k proofs of one circuit in one launch sequence (zk_prove_batch): list of proof JSON strings, each byte-identical to
    prove(ctx, witness_p)"""
    w = _c64(witnesses).reshape(-1, ctx.r1cs.V + 1, 4)
    return [proof_to_json(p, w[i, 1:1 + ctx.r1cs.nIn], canonical) for i, p in enumerate(ctx.prove_batch_structs(w, canonical))]


def prove(ctx, witness, canonical=False):
    """ethsnarks::prove(ProverContextT&, ProtoboardT&) (src/stubs.cpp:42-47): proof JSON string.
    `witness` plays pb.values (ONE at index 0); primary input = witness[1..nIn]."""
    w = _c64(witness).reshape(-1, 4)
    proof = ctx.prove_struct(w, canonical)
    return proof_to_json(proof, w[1:1 + ctx.r1cs.nIn], canonical)


def stub_prove_from_pb(r1cs, witness, pk_file, **kw):
    """upstream ethsnarks convenience wrapper still referenced by src/pinocchio/main.cpp:10,41."""
    pk = load_proving_key(pk_file)
    ctx = ProverContext(pk, r1cs, **kw)
    try:
        return prove(ctx, witness)
    finally:
        ctx.close()
        pk.close()


def stub_verify(vk_json, proof_json):
    """ethsnarks::stub_verify (src/stubs.cpp:16-33): True iff the proof verifies under the key (host code)."""
    ok = C.c_int(0)
    _check(load_library(_lib_path_loaded).zk_verify(vk_json.encode(), proof_json.encode(), C.byref(ok)))
    return ok.value == 1


def stub_test_proof_verify(r1cs, witness, **kw):
    """ethsnarks::stub_test_proof_verify (src/stubs.cpp:135-148): keygen -> prove -> verify in memory,
    with the context fully initialised (the reference forgets constraint_system and domain)."""
    pk, vk = keygen(r1cs, **kw)
    ctx = ProverContext(pk, r1cs)
    try:
        return stub_verify(vk.to_json(), prove(ctx, witness))
    finally:
        ctx.close()


# ---- witness completion on the GPU
class PinnedBuffer:
    """zk_host_alloc: pinned host memory the H2D copy of zk_prove_submit_pinned reads in place; `array` is a numpy uint64 view"""

    def __init__(self, nbytes):
        lib = load_library(_lib_path_loaded)
        p = C.c_void_p()
        _check(lib.zk_host_alloc(C.c_size_t(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes
        self.array = np.frombuffer((C.c_uint8 * nbytes).from_address(self.ptr), dtype=np.uint64)

    def free(self):
        if self.ptr:
            self.array = None
            load_library(_lib_path_loaded).zk_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """device memory through the library (zk_dev_*): for hosts without a HIP binding of their own"""

    def __init__(self, nbytes, device=0):
        p = C.c_void_p()
        _check(load_library(_lib_path_loaded).zk_dev_alloc(C.c_size_t(nbytes), device, C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        _check(_lib.zk_dev_upload(C.c_void_p(self.ptr + offset), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)))

    def download(self, shape, dtype=np.uint64, offset=0):
        out = np.zeros(shape, dtype=dtype)
        _check(_lib.zk_dev_download(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr and _lib is not None:
            _lib.zk_dev_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class WitnessPlan:
    """zk_wplan: the constraint system as a witness program (forward substitution), run on the GPU for k witnesses at a time.
    `known`: indices of the variables the caller supplies (ONE is implied)."""

    def __init__(self, r1cs, known, device=0, bit_hints=(), inv_hints=(), nonzero_hints=()):
        """bit_hints: (src, first, count) triples -- w[first + i] = bit i of w[src] (ZK_WHINT_BITS: advice the constraints only check);
        inv_hints: (src, dst) pairs -- w[dst] = 1 / w[src], 0 for 0 (ZK_WHINT_INV); nonzero_hints: (src, dst) -- w[dst] = [w[src] != 0]
        (ZK_WHINT_NONZERO): the M and Y of the reference's IsNonZero gadget, src/gadgets/isnonzero.cpp:48-60"""
        lib = load_library(_lib_path_loaded)
        flags = np.zeros(r1cs.V + 1, dtype=np.uint8)
        flags[np.asarray(list(known), dtype=np.int64)] = 1
        keep = []
        a, b, c = _csr_structs(r1cs, keep)
        h = C.c_void_p()
        hints = np.asarray([[1, s_, f_, n_] for s_, f_, n_ in bit_hints] + [[2, s_, d_, 1] for s_, d_ in inv_hints] + [[3, s_, d_, 1] for s_, d_ in nonzero_hints], dtype=np.uint32).reshape(-1, 4)
        _check(lib.zk_wplan_create_hinted(C.byref(a), C.byref(b), C.byref(c), C.c_uint32(r1cs.nC), C.c_uint32(r1cs.V),
                                          flags.ctypes.data_as(C.POINTER(C.c_uint8)),
                                          hints.ctypes.data_as(C.c_void_p) if len(hints) else None, C.c_uint32(len(hints)), device, C.byref(h)))
        self._h, self.r1cs = h, r1cs

    def solve(self, device_ptr, k):
        """complete k witnesses in place (device memory); returns the number of violated check constraints"""
        bad = C.c_uint32(0)
        _check(_lib.zk_wplan_solve(self._h, C.c_void_p(device_ptr), C.c_uint32(k), C.byref(bad)))
        return int(bad.value)

    def close(self):
        if getattr(self, "_h", None) is not None and _lib is not None:
            _lib.zk_wplan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- measurement aids
def launch_count():
    return int(load_library(_lib_path_loaded).zk_launch_count())


def profile_begin():
    _check(load_library(_lib_path_loaded).zk_profile_begin())


def profile_end():
    """(sum of kernel durations in ms, launches, {kernel: (calls, ms)})"""
    s, n = C.c_float(0), C.c_uint32(0)
    buf = C.create_string_buffer(1 << 16)
    _check(load_library(_lib_path_loaded).zk_profile_end(C.byref(s), C.byref(n), buf, C.c_size_t(len(buf))))
    per = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.rsplit(" ", 2)
        per[name] = (int(calls), float(ms))
    return float(s.value), int(n.value), per


def device_info(device=0):
    cu, mhz = C.c_uint32(0), C.c_uint32(0)
    name = C.create_string_buffer(256)
    _check(load_library(_lib_path_loaded).zk_device_info(device, C.byref(cu), C.byref(mhz), name, C.c_size_t(256)))
    return {"compute_units": int(cu.value), "clock_mhz": int(mhz.value), "name": name.value.decode()}


# ---- kernel-level entry points
def device_pci_bus_id(device=0):
    buf = C.create_string_buffer(64)
    _check(load_library(_lib_path_loaded).zk_device_pci_bus_id(int(device), buf, C.c_size_t(len(buf))))
    return buf.value.decode()


def ntt(data, logm, inverse=False, coset=False, device=0):
    a = _c64(data).copy()
    _check(load_library(_lib_path_loaded).zk_ntt(_p64(a), C.c_uint32(logm), int(inverse), int(coset), device))
    return a


def msm(bases, scalars, g2=False, c=0, device=0):
    lib = load_library(_lib_path_loaded)
    bases = _c64(bases); scalars = _c64(scalars)
    n = scalars.size // 4
    out = np.zeros(16 if g2 else 8, dtype=np.uint64)
    fn = lib.zk_msm_g2 if g2 else lib.zk_msm_g1
    _check(fn(_p64(bases) if n else None, _p64(scalars) if n else None, C.c_uint32(n), C.c_uint32(c), device, _p64(out)))
    return out


def field_mul(a, b, field="fr", device=0):
    a = _c64(a); b = _c64(b)
    out = np.zeros_like(a)
    _check(load_library(_lib_path_loaded).zk_field_mul(_p64(a), _p64(b), _p64(out), C.c_uint32(a.size // 4), 0 if field == "fr" else 1, device))
    return out
