"""R1CS containers (CSR) and the synthetic circuits the measurement configs use.

The reference hands the prover a `libsnark::r1cs_constraint_system` whose rows are read through
`constraints[c]->getA()/getB()/getC()` -> `getTerms()` -> `{index, coeff}` (src/export.cpp:157-190);
the C ABI takes the same information flattened once into CSR (three matrices, variable index 0 is
the constant ONE, coefficients Montgomery Fr).

`synthetic_chain` restates the shape of libsnark's `generate_r1cs_example_with_field_input`
(used by src/r1cs_gg_ppzksnark_zok/profiling/profile_r1cs_gg_ppzksnark_zok.cpp:64-66; body ABSENT
from the checkout) with a SplitMix64-seeded start, as fixed in SURVEY.md section 8(d).
"""
from dataclasses import dataclass
import numpy as np
from .fields import FR, MONT_R, ints_to_limbs, fr_to_mont, FR_ONE_MONT

SEED_DEFAULT = 0x657468736E61726B  # "ethsnark"
_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.x = seed & _M64

    def next(self):
        self.x = (self.x + 0x9E3779B97F4A7C15) & _M64
        z = self.x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def fr(self):
        """4 draws -> 256 bits (first draw = least significant limb) -> mod r."""
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % FR


@dataclass
class CSR:
    row_ptr: np.ndarray   # uint32 [n_rows + 1]
    col: np.ndarray       # uint32 [nnz]
    coeff: np.ndarray     # uint64 [nnz, 4], Montgomery Fr

    @property
    def n_rows(self):
        return len(self.row_ptr) - 1

    @property
    def nnz(self):
        return len(self.col)

    @staticmethod
    def from_rows(rows):
        """rows: list of lists of (var_index, int coeff)."""
        ptr = np.zeros(len(rows) + 1, dtype=np.uint32)
        cols, coefs = [], []
        for j, row in enumerate(rows):
            for i, c in row:
                cols.append(i)
                coefs.append(c % FR)
            ptr[j + 1] = len(cols)
        coeff = fr_to_mont(coefs) if coefs else np.zeros((0, 4), dtype=np.uint64)
        return CSR(ptr, np.asarray(cols, dtype=np.uint32), coeff)

    def to_rows(self):
        from .fields import fr_from_mont
        co = fr_from_mont(self.coeff) if self.nnz else []
        return [[(int(self.col[k]), co[k]) for k in range(self.row_ptr[j], self.row_ptr[j + 1])]
                for j in range(self.n_rows)]


@dataclass
class R1CS:
    nC: int
    nIn: int
    V: int          # variables excluding the constant ONE; witness has V + 1 entries
    A: CSR
    B: CSR
    C: CSR

    @property
    def domain_size(self):
        """get_domain, src/stubs.cpp:49-65 (32-bit roundUpToNearestPowerOf2)."""
        v = self.nC + self.nIn + 1
        m = 1
        while m < v:
            m <<= 1
        return m

    @property
    def nnz(self):
        return self.A.nnz + self.B.nnz + self.C.nnz

    def is_satisfied(self, w_ints):
        for ra, rb, rc in zip(self.A.to_rows(), self.B.to_rows(), self.C.to_rows()):
            dot = lambda r: sum(c * w_ints[i] for i, c in r) % FR
            if dot(ra) * dot(rb) % FR != dot(rc):
                return False
        return True

    def as_pyref(self):
        return (self.nC, self.nIn, self.V, self.A.to_rows(), self.B.to_rows(), self.C.to_rows())


def synthetic_chain(nC, nIn=1, seed=SEED_DEFAULT):
    """Chain circuit of SURVEY 8(d): V = nC + 2, rows i = 0..nC-2 alternate
    even: (x_{i+1} + x_{i+2}) * 1 = x_{i+3},  odd: x_{i+1} * x_{i+2} = x_{i+3};
    last row: (sum_{i<V} x_i) * (sum_{i<V} x_i) = x_V  (dense, 2(V-1) terms).
    Returns (R1CS, witness ints [V+1])."""
    assert nC >= 2
    V = nC + 2
    rng = SplitMix64(seed)
    w = [1, rng.fr(), rng.fr()]
    for i in range(nC - 1):
        a, b = w[i + 1], w[i + 2]
        w.append((a + b) % FR if i % 2 == 0 else a * b % FR)
    s = sum(w[1:V]) % FR
    w.append(s * s % FR)
    assert len(w) == V + 1
    i = np.arange(nC - 1, dtype=np.int64)
    even = (i % 2 == 0)
    # A: even rows two terms (x_{i+1}, x_{i+2}), odd rows one term (x_{i+1}); last row V-1 terms
    a_cnt = np.where(even, 2, 1)
    a_ptr = np.zeros(nC + 1, dtype=np.int64)
    a_ptr[1:nC] = np.cumsum(a_cnt)
    a_ptr[nC] = a_ptr[nC - 1] + (V - 1)
    a_col = np.empty(a_ptr[nC], dtype=np.uint32)
    a_col[a_ptr[:nC - 1]] = i + 1
    a_col[a_ptr[:nC - 1][even] + 1] = i[even] + 2
    a_col[a_ptr[nC - 1]:] = np.arange(1, V, dtype=np.uint32)
    # B: even rows ONE, odd rows x_{i+2}; last row V-1 terms
    b_ptr = np.zeros(nC + 1, dtype=np.int64)
    b_ptr[1:nC] = np.arange(1, nC)
    b_ptr[nC] = b_ptr[nC - 1] + (V - 1)
    b_col = np.empty(b_ptr[nC], dtype=np.uint32)
    b_col[:nC - 1] = np.where(even, 0, i + 2)
    b_col[nC - 1:] = np.arange(1, V, dtype=np.uint32)
    # C: x_{i+3}; last row x_V
    c_ptr = np.arange(nC + 1, dtype=np.int64)
    c_col = np.empty(nC, dtype=np.uint32)
    c_col[:nC - 1] = i + 3
    c_col[nC - 1] = V
    ones = lambda n: np.broadcast_to(FR_ONE_MONT, (n, 4)).copy()
    mk = lambda p, c: CSR(p.astype(np.uint32), c, ones(len(c)))
    return R1CS(nC, nIn, V, mk(a_ptr, a_col), mk(b_ptr, b_col), mk(c_ptr, c_col)), w


def random_r1cs(nC, nIn, n_extra_vars=3, max_terms=4, seed=1, small_values=False, witness_seed=None):
    """Random satisfiable R1CS with general coefficients (parity-test material):
    row j: <A_j, x> * <B_j, x> = k_j * x_new; some rows are zero rows; `n_extra_vars` variables are
    never referenced (their L-query entries are the point at infinity, A/B entries absent).
    small_values: witness seeds drawn from {0, 1, 2, 3} to exercise the 0/1 scalar partitions.
    witness_seed: the same constraint system (it depends on `seed` alone) with another satisfying witness."""
    rng = SplitMix64(seed)
    wrng = SplitMix64(witness_seed) if witness_seed is not None else None
    n_free = nIn + 2
    w = [1] + [(rng.next() % 4 if small_values else rng.fr()) for _ in range(n_free)]
    if wrng is not None:
        w = [1] + [(wrng.next() % 4 if small_values else wrng.fr()) for _ in range(n_free)]
    A, B, C = [], [], []
    for j in range(nC):
        if j % 7 == 5:                       # zero row: 0 * 0 = 0
            A.append([]); B.append([]); C.append([])
            continue
        nv = len(w)
        ra = [(rng.next() % nv, (rng.next() % 5 if small_values else rng.fr())) for _ in range(1 + rng.next() % max_terms)]
        rb = [(rng.next() % nv, (rng.next() % 5 if small_values else rng.fr())) for _ in range(1 + rng.next() % max_terms)]
        dot = lambda r: sum(c * w[i] for i, c in r) % FR
        k = 1 if (small_values or j % 3 == 0) else (rng.fr() or 1)
        val = dot(ra) * dot(rb) % FR * pow(k, -1, FR) % FR
        w.append(val)
        A.append(ra); B.append(rb); C.append([(nv, k)])
    for _ in range(n_extra_vars):
        v = rng.fr()
        w.append(wrng.fr() if wrng is not None else v)
    V = len(w) - 1
    return R1CS(nC, nIn, V, CSR.from_rows(A), CSR.from_rows(B), CSR.from_rows(C)), w


# --------------------------------------------------------------------------------------------------
# The reference's circuit / witness dumps (src/export.cpp:157-221): an existing ethsnarks binary can write its
# protoboard with r1cs2json / witness2json and this backend proves it without linking libsnark.
def r1cs_to_json(r):
    """exact text of r1cs2json (src/export.cpp:173-206); nVars counts the constant ONE"""
    def lc(row):
        return "{" + ",".join('"%d": "%d"' % (i, c) for i, c in row) + "}"
    A, B, C = r.A.to_rows(), r.B.to_rows(), r.C.to_rows()
    s = "{\n \"nPubInputs\": %d,\n \"nOutputs\": 0,\n \"nVars\": %d,\n \"nConstraints\": %d,\n \"constraints\": [\n" % (r.nIn, r.V + 1, r.nC)
    for j in range(r.nC):
        s += "  [" + lc(A[j]) + "," + lc(B[j]) + "," + lc(C[j]) + ("]\n" if j == r.nC - 1 else "],\n")
    return s + " ]\n}"


def r1cs_from_json(text):
    import json
    d = json.loads(text)
    rows = lambda k: [[(int(i), int(c)) for i, c in con[k].items()] for con in d["constraints"]]
    nC = int(d["nConstraints"])
    assert nC == len(d["constraints"])
    return R1CS(nC, int(d["nPubInputs"]), int(d["nVars"]) - 1, CSR.from_rows(rows(0)), CSR.from_rows(rows(1)), CSR.from_rows(rows(2)))


def witness_to_json(w_ints):
    """exact text of witness2json (src/export.cpp:208-221): decimal strings, index 0 is ONE"""
    return "[\n" + ",\n".join(' "%d"' % v for v in w_ints) + "\n]"


def witness_from_json(text):
    import json
    return [int(v) % FR for v in json.loads(text)]
