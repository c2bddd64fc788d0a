"""alt_bn128 field constants and int <-> limb-array conversions for the host side.

Memory form everywhere (host buffers, .raw key stream, device arrays): 4 x u64 little-endian limbs,
Montgomery representation with R = 2^256 -- the in-memory form of libff::Fp_model<4> that
`pb.values` and the `.raw` proving key hold in the reference (CMakeLists.txt:115-131).
Moduli: contracts/Verifier.sol:10,17.
"""
import numpy as np

FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
FQ = 21888242871839275222246405745257275088696311157297823662689037894645226208583
MONT_R = 1 << 256
FR_RINV = pow(MONT_R, -1, FR)
FQ_RINV = pow(MONT_R, -1, FQ)


def ints_to_limbs(vals):
    """list of python ints (< 2^256) -> uint64 array (n, 4), little-endian limbs."""
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u8").reshape(-1, 4).copy()


def limbs_to_ints(arr):
    a = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, 4)
    raw = a.tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(a.shape[0])]


def fr_to_mont(vals):
    return ints_to_limbs([(int(v) % FR) * MONT_R % FR for v in vals])


def fr_from_mont(arr):
    return [v * FR_RINV % FR for v in limbs_to_ints(arr)]


def fq_to_mont(vals):
    return ints_to_limbs([(int(v) % FQ) * MONT_R % FQ for v in vals])


def fq_from_mont(arr):
    return [v * FQ_RINV % FQ for v in limbs_to_ints(arr)]


FR_ONE_MONT = ints_to_limbs([MONT_R % FR])[0]
