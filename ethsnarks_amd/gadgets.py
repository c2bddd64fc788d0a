"""R1CS / witness front end for the two real ethsnarks circuits the measurement configs name
(BASELINE configs 1 and 4): MiMC-e7 (Miyaguchi-Preneel hash) and the Merkle path
authenticator.  It restates the *constraint generators* of the reference gadgets so that those circuits can
be proven on a box that has neither libsnark nor the reference checkout:

    MiMCe7_round / MiMC_gadget             src/gadgets/mimc.hpp:115-318   (x^7 rounds, Keccak-256 round constants)
    MiyaguchiPreneel_OWF                   src/gadgets/onewayfunction.hpp:67-127
    merkle_path_selector                   src/gadgets/merkle_tree.cpp:11-63
    markle_path_compute / _authenticator   src/gadgets/merkle_tree.hpp:71-191
    merkle_tree_IVs                        src/gadgets/merkle_tree.cpp:75-113 (= ethsnarks/merkletree.py:36-44)

Variable allocation order and constraint order follow the C++ constructors / generate_r1cs_constraints, so
the constraint system has the reference's shape (21 345 constraints at depth 29).  The native (out of
circuit) functions mirror ethsnarks/mimc/permutation.py and ethsnarks/merkletree.py and are checked against
the reference's known answers in tests/test_gadgets.py.  This is circuit *authoring* glue (host side,
Python, off the proving path); the prover only ever sees the CSR + witness it produces.
"""
import hashlib
from .fields import FR
from .r1cs import CSR, R1CS

# ----------------------------------------------------------------------------- Keccak-256 (original padding 0x01)
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
       0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
       0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
       0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _M if n else v


def _keccak_f(A):
    for rc in _RC:
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= rc
    return A


def keccak256(data):
    rate = 136
    p = bytearray(data)
    p.append(0x01)
    while len(p) % rate:
        p.append(0)
    p[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(p), rate):
        blk = p[off:off + rate]
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(blk[8 * i:8 * i + 8], "little")
        A = _keccak_f(A)
    return b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


# ----------------------------------------------------------------------------- native MiMC / Merkle (permutation.py, merkletree.py)
def H(*args):
    data = b"".join(a if isinstance(a, bytes) else int(a).to_bytes(32, "big") for a in args)
    return int.from_bytes(keccak256(data), "big")


def mimc_constants(seed=b"mimc", rounds=91):
    s = H(seed)
    out = []
    for _ in range(rounds):
        s = H(s)
        out.append(s % FR)        # the gadget reduces mod r (libff Fp(bigint)); the hash chain itself is unreduced
    return out


_MIMC_C = None


def _consts():
    global _MIMC_C
    if _MIMC_C is None:
        _MIMC_C = mimc_constants()
    return _MIMC_C


def mimc(x, k, e=7):
    for c in _consts():
        x = pow((x + k + c) % FR, e, FR)
    return (x + k) % FR


def mimc_hash(msgs, iv=0):
    """Miyaguchi-Preneel: k_{i+1} = k_i + E_{k_i}(m_i) + m_i"""
    k = iv
    for m in msgs:
        k = (k + mimc(m, k) + m) % FR
    return k


def merkle_ivs(depth=29):
    out, h = [], hashlib.sha256()
    for i in range(depth):
        h.update(b"MerkleTree-" + int(i).to_bytes(2, "little"))
        out.append(int.from_bytes(h.digest(), "big") % FR)
    return out


def merkle_unique(depth, index):
    return int.from_bytes(hashlib.sha256(int(depth).to_bytes(2, "big") + int(index).to_bytes(30, "big")).digest(), "big") % FR


def merkle_root(leaf, address_bits, path, ivs):
    item = leaf
    for d, (bit, sib) in enumerate(zip(address_bits, path)):
        item = mimc_hash([sib, item] if bit else [item, sib], ivs[d])
    return item


# ----------------------------------------------------------------------------- a minimal protoboard
class Protoboard:
    """libsnark::protoboard in miniature: variables are 1-based, index 0 is the constant ONE."""

    def __init__(self):
        self.values = [1]
        self.A, self.B, self.C = [], [], []
        self.n_inputs = 0

    def allocate(self, value=0):
        self.values.append(value % FR)
        return len(self.values) - 1

    def allocate_array(self, n, values=None):
        return [self.allocate(values[i] if values else 0) for i in range(n)]

    def val(self, var):
        return self.values[var]

    def set_val(self, var, v):
        self.values[var] = v % FR

    def set_input_sizes(self, n):
        self.n_inputs = n

    @staticmethod
    def _lc(x):
        """linear combination: int constant | variable index wrapped as ('v', idx) | dict idx->coeff"""
        if isinstance(x, dict):
            return {i: c % FR for i, c in x.items() if c % FR}
        if isinstance(x, tuple):
            return {x[1]: 1}
        return {0: x % FR} if x % FR else {}

    def add_r1cs_constraint(self, a, b, c):
        self.A.append(self._lc(a)); self.B.append(self._lc(b)); self.C.append(self._lc(c))

    def num_constraints(self):
        return len(self.A)

    def is_satisfied(self):
        dot = lambda lc: sum(c * self.values[i] for i, c in lc.items()) % FR
        return all(dot(a) * dot(b) % FR == dot(c) for a, b, c in zip(self.A, self.B, self.C))

    def to_r1cs(self):
        rows = lambda M: [sorted(lc.items()) for lc in M]
        return R1CS(len(self.A), self.n_inputs, len(self.values) - 1,
                    CSR.from_rows(rows(self.A)), CSR.from_rows(rows(self.B)), CSR.from_rows(rows(self.C))), list(self.values)


def V(i):
    return ("v", i)


def lc_add(*terms):
    """sum of linear combinations / variables / constants -> dict"""
    out = {}
    for t in terms:
        for i, c in Protoboard._lc(t).items():
            out[i] = (out.get(i, 0) + c) % FR
    return out


def lc_scale(t, k):
    return {i: c * k % FR for i, c in Protoboard._lc(t).items()}


# ----------------------------------------------------------------------------- gadgets
class MiMCe7Round:
    """src/gadgets/mimc.hpp:115-183"""

    def __init__(self, pb, x, k, C, add_k_to_result):
        self.pb, self.x, self.k, self.C, self.add_k = pb, x, k, C, add_k_to_result
        self.a, self.b, self.c, self.d = pb.allocate(), pb.allocate(), pb.allocate(), pb.allocate()

    def result(self):
        return self.d

    def generate_r1cs_constraints(self):
        pb = self.pb
        t = lc_add(V(self.x), V(self.k), self.C)
        pb.add_r1cs_constraint(t, t, V(self.a))
        pb.add_r1cs_constraint(V(self.a), V(self.a), V(self.b))
        pb.add_r1cs_constraint(V(self.a), V(self.b), V(self.c))
        pb.add_r1cs_constraint(t, V(self.c), lc_add(V(self.d), lc_scale(V(self.k), -1)) if self.add_k else V(self.d))

    def generate_r1cs_witness(self):
        pb = self.pb
        k = pb.val(self.k)
        t = (pb.val(self.x) + k + self.C) % FR
        a = t * t % FR; b = a * a % FR; c = a * b % FR
        pb.set_val(self.a, a); pb.set_val(self.b, b); pb.set_val(self.c, c)
        pb.set_val(self.d, c * t + (k if self.add_k else 0))


class MiMCe7Gadget:
    """MiMC_gadget<MiMCe7_round>, src/gadgets/mimc.hpp:186-275"""

    def __init__(self, pb, x, k):
        self.rounds = []
        cs = _consts()
        for i, C in enumerate(cs):
            rx = x if i == 0 else self.rounds[-1].result()
            self.rounds.append(MiMCe7Round(pb, rx, k, C, i == len(cs) - 1))

    def result(self):
        return self.rounds[-1].result()

    def generate_r1cs_constraints(self):
        for r in self.rounds:
            r.generate_r1cs_constraints()

    def generate_r1cs_witness(self):
        for r in self.rounds:
            r.generate_r1cs_witness()


class MiMCe7HashGadget:
    """MiyaguchiPreneel_OWF<MiMC_e7_gadget>, src/gadgets/onewayfunction.hpp:67-127"""

    def __init__(self, pb, iv, messages):
        self.pb, self.iv, self.messages = pb, iv, list(messages)
        self.outputs = pb.allocate_array(len(self.messages))
        self.ciphers = []
        for i, m in enumerate(self.messages):
            key = iv if i == 0 else self.outputs[i - 1]
            self.ciphers.append(MiMCe7Gadget(pb, m, key))

    def result(self):
        return self.outputs[-1]

    def generate_r1cs_constraints(self):
        for i, c in enumerate(self.ciphers):
            c.generate_r1cs_constraints()
            key = self.iv if i == 0 else self.outputs[i - 1]
            self.pb.add_r1cs_constraint(lc_add(V(key), V(c.result()), V(self.messages[i])), 1, V(self.outputs[i]))

    def generate_r1cs_witness(self):
        for i, c in enumerate(self.ciphers):
            c.generate_r1cs_witness()
            key = self.pb.val(self.iv if i == 0 else self.outputs[i - 1])
            self.pb.set_val(self.outputs[i], key + self.pb.val(c.result()) + self.pb.val(self.messages[i]))


class MerklePathSelector:
    """src/gadgets/merkle_tree.cpp:11-63"""

    def __init__(self, pb, inp, pathvar, is_right):
        self.pb, self.inp, self.pathvar, self.is_right = pb, inp, pathvar, is_right
        self.left_a, self.left_b, self.left = pb.allocate(), pb.allocate(), pb.allocate()
        self.right_a, self.right_b, self.right = pb.allocate(), pb.allocate(), pb.allocate()

    def generate_r1cs_constraints(self):
        pb = self.pb
        not_right = lc_add(1, lc_scale(V(self.is_right), -1))
        pb.add_r1cs_constraint(not_right, V(self.inp), V(self.left_a))
        pb.add_r1cs_constraint(V(self.is_right), V(self.pathvar), V(self.left_b))
        pb.add_r1cs_constraint(lc_add(V(self.left_a), V(self.left_b)), 1, V(self.left))
        pb.add_r1cs_constraint(V(self.is_right), V(self.inp), V(self.right_a))
        pb.add_r1cs_constraint(not_right, V(self.pathvar), V(self.right_b))
        pb.add_r1cs_constraint(lc_add(V(self.right_a), V(self.right_b)), 1, V(self.right))

    def generate_r1cs_witness(self):
        pb = self.pb
        r, i, p = pb.val(self.is_right), pb.val(self.inp), pb.val(self.pathvar)
        pb.set_val(self.left_a, (1 - r) * i); pb.set_val(self.left_b, r * p)
        pb.set_val(self.left, pb.val(self.left_a) + pb.val(self.left_b))
        pb.set_val(self.right_a, r * i); pb.set_val(self.right_b, (1 - r) * p)
        pb.set_val(self.right, pb.val(self.right_a) + pb.val(self.right_b))


class MerklePathAuthenticator:
    """merkle_path_authenticator<MiMC_e7_hash_gadget>, src/gadgets/merkle_tree.hpp:71-191"""

    def __init__(self, pb, depth, address_bits, ivs, leaf, expected_root, path):
        assert depth > 0 and len(address_bits) == depth and len(ivs) >= depth
        self.pb, self.expected_root = pb, expected_root
        self.selectors, self.hashers = [], []
        for i in range(depth):
            inp = leaf if i == 0 else self.hashers[i - 1].result()
            sel = MerklePathSelector(pb, inp, path[i], address_bits[i])
            self.selectors.append(sel)
            self.hashers.append(MiMCe7HashGadget(pb, ivs[i], [sel.left, sel.right]))

    def result(self):
        return self.hashers[-1].result()

    def generate_r1cs_constraints(self):
        for s, h in zip(self.selectors, self.hashers):
            s.generate_r1cs_constraints()
            h.generate_r1cs_constraints()
        self.pb.add_r1cs_constraint(V(self.result()), 1, V(self.expected_root))

    def generate_r1cs_witness(self):
        for s, h in zip(self.selectors, self.hashers):
            s.generate_r1cs_witness()
            h.generate_r1cs_witness()

    def is_valid(self):
        return self.pb.val(self.result()) == self.pb.val(self.expected_root)


# ----------------------------------------------------------------------------- the two measurement circuits
def merkle_membership_circuit(depth=29, leaf=None, address=0, path=None, root_public=True):
    """BASELINE config 4: merkle_path_authenticator<MiMC_e7_hash_gadget> at `depth` (29 -> 21 345 constraints,
    domain 2^15).  Default witness = the reference's depth-29 known-answer tree (test/test_merkle.py:82-107):
    leaf 0 = item_a, its sibling item_b, the other siblings the `unique` placeholders.  The root is the single
    public input (the in-tree C++ test declares none, src/test/test_merkle_tree.cpp:56-110)."""
    item_a = 3703141493535563179657531719960160174296085208671919316200479060314459804651
    item_b = 134551314051432487569247388144051420116740427803855572138106146683954151557
    if leaf is None:
        leaf = item_a
    if path is None:
        path = [item_b] + [merkle_unique(d, 1) for d in range(1, depth)]
    bits = [(address >> i) & 1 for i in range(depth)]
    ivs_val = merkle_ivs(29)
    root = merkle_root(leaf, bits, path, ivs_val)
    pb = Protoboard()
    expected_root = pb.allocate(root)
    if root_public:
        pb.set_input_sizes(1)
    address_bits = pb.allocate_array(depth, bits)
    path_vars = pb.allocate_array(depth, path)
    leaf_var = pb.allocate(leaf)
    ivs = pb.allocate_array(29, ivs_val)                          # merkle_tree_IVs(pb): 29 variables holding constants
    auth = MerklePathAuthenticator(pb, depth, address_bits, ivs, leaf_var, expected_root, path_vars)
    auth.generate_r1cs_witness()
    auth.generate_r1cs_constraints()
    assert auth.is_valid()
    r1cs, w = pb.to_r1cs()
    return r1cs, w, root


def mimc_preimage_circuit(n_words=11, seed=7):
    """BASELINE config 1 (in-tree stand-in for the SHA256 'hashpreimage'): MiMC-e7 Miyaguchi-Preneel hash over
    n_words message words, the digest public: n_words * (91*4 + 1) constraints (4 015 at 11 words -> domain 2^12)."""
    from .r1cs import SplitMix64
    rng = SplitMix64(seed)
    msgs = [rng.fr() for _ in range(n_words)]
    digest = mimc_hash(msgs, 0)
    pb = Protoboard()
    out = pb.allocate(digest)
    pb.set_input_sizes(1)
    iv = pb.allocate(0)
    m = pb.allocate_array(n_words, msgs)
    g = MiMCe7HashGadget(pb, iv, m)
    g.generate_r1cs_witness()
    g.generate_r1cs_constraints()
    pb.add_r1cs_constraint(V(g.result()), 1, V(out))
    r1cs, w = pb.to_r1cs()
    return r1cs, w, digest


def field2bits_circuit(n_bits=253, value=None, seed=5):
    """A circuit with NON-DETERMINISTIC advice, shaped like the reference's field2bits gadgets (src/gadgets/field2bits_strict.cpp without
    the strict range comparison): public x, bits b_0 .. b_{n-1} with  b_i (1 - b_i) = 0  and  (sum 2^i b_i) * 1 = x,  then the bits are USED:
    lo = b_0 + 2 b_1 + 4 b_2,  y = lo * x,  digest = MiMC-e7 hash of (y, x).  The constraint system alone does not tell a forward
    substitution what the bits are -- zk_wplan needs the ZK_WHINT_BITS hint (src = x, first = b_0, count = n_bits).
    Returns (R1CS, witness, (x_var, first_bit_var, n_bits))."""
    from .r1cs import SplitMix64
    if value is None:
        value = SplitMix64(seed).fr() % (1 << n_bits)
    assert 0 <= value < (1 << n_bits) and (1 << n_bits) <= FR
    pb = Protoboard()
    x = pb.allocate(value)
    pb.set_input_sizes(1)
    bits = pb.allocate_array(n_bits, [(value >> i) & 1 for i in range(n_bits)])
    for b in bits:
        pb.add_r1cs_constraint(V(b), lc_add(1, lc_scale(V(b), FR - 1)), 0)          # b (1 - b) = 0
    pb.add_r1cs_constraint({b: (1 << i) % FR for i, b in enumerate(bits)}, 1, V(x))     # the bits spell x
    lo = (value & 7) if n_bits >= 3 else value
    y = pb.allocate(lo * value % FR)
    pb.add_r1cs_constraint({b: 1 << i for i, b in enumerate(bits[:3])}, V(x), V(y))
    iv = pb.allocate(0)
    pb.add_r1cs_constraint(V(iv), 1, 0)                                               # iv = 0 (a checked constant)
    g = MiMCe7HashGadget(pb, iv, [y, x])
    g.generate_r1cs_witness()
    g.generate_r1cs_constraints()
    r1cs, w = pb.to_r1cs()
    assert pb.is_satisfied()
    return r1cs, w, (x, bits[0], n_bits)


def isnonzero_circuit(values=(0, 5, 0, 7, 1), seed=11):
    """The reference's IsNonZero gadget (src/gadgets/isnonzero.cpp:34-60) over a vector of private values, its results used: for every X the
    constraints  Y (1 - Y) = 0,  X (1 - Y) = 0,  X M = Y  with the advice M = 1 / X (0 for X = 0) and Y = [X != 0] (generate_r1cs_witness,
    :48-60), then  count = sum Y  (public),  t = count * first X,  digest = MiMC-e7 hash of (t, count).  The constraints only CHECK M and Y, and the
    first one reads Y before anything defines it: zk_wplan needs ZK_WHINT_INV (src = X, dst = M) and ZK_WHINT_NONZERO (src = X, dst = Y).
    Returns (R1CS, witness, [(x_var, y_var, m_var), ...])."""
    pb = Protoboard()
    ys = [1 if v % FR else 0 for v in values]
    count = pb.allocate(sum(ys))
    pb.set_input_sizes(1)
    triples = []
    for v, yv in zip(values, ys):
        x = pb.allocate(v % FR)
        y = pb.allocate(yv)
        m = pb.allocate(pow(v % FR, FR - 2, FR))                                       # 0 for 0
        pb.add_r1cs_constraint(V(y), lc_add(1, lc_scale(V(y), FR - 1)), 0)             # generate_boolean_r1cs_constraint(Y)
        pb.add_r1cs_constraint(V(x), lc_add(1, lc_scale(V(y), FR - 1)), 0)             # X (1 - Y) = 0
        pb.add_r1cs_constraint(V(x), V(m), V(y))                                       # X (1/X) = Y
        triples.append((x, y, m))
    pb.add_r1cs_constraint({y: 1 for _, y, _ in triples}, 1, V(count))                 # the results are used
    t = pb.allocate(sum(ys) * (values[0] % FR) % FR)
    pb.add_r1cs_constraint(V(count), V(triples[0][0]), V(t))
    iv = pb.allocate(0)
    pb.add_r1cs_constraint(V(iv), 1, 0)
    g = MiMCe7HashGadget(pb, iv, [t, count])
    g.generate_r1cs_witness()
    g.generate_r1cs_constraints()
    r1cs, w = pb.to_r1cs()
    assert pb.is_satisfied()
    return r1cs, w, triples

