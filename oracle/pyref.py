"""
oracle/pyref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent pure-Python big-int restatement of the Groth16 "no-ZK, context" proving path of
zkh2018/ethsnarks plus a pairing verifier.  It exists so that the C oracle (oracle/*.c) and, through
it, the HIP backend can be pinned:

  * verifier:  validated against the reference's only cross-implementation vector, the static
               (vk, proof, inputs) triple of test/test_verify.py:10-12 (== test/TestVerifier.sol:11-55),
               committed as tests/golden/ref_static_triple.json.
  * prover:    restates src/r1cs_gg_ppzksnark_zok/r1cs_gg_ppzksnark_zok.tcc:451-550 with the
               witness map of SURVEY.md Appendix A.3 and the JSON writer of src/export.cpp:20-121.
  * keygen:    restates ...tcc:277-449 (+ nozk conversion ...hpp:209-233) with seeded toxic waste.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Everything here is slow by design (affine big-int arithmetic); use it at m <= 2^8.
"""

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr, contracts/Verifier.sol:10
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq, contracts/Verifier.sol:17
ROOT_2_28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904  # 5^((r-1)/2^28)
COSET_G = 5
G1_GEN = (1, 2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))
ATE_LOOP = 29793968203157093288  # 6z+2, z = 4965661367192848881


# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1), tuples (c0, c1)
def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2_muls(a, s): return ((a[0] * s) % Q, (a[1] * s) % Q)
def f2_conj(a): return (a[0], (-a[1]) % Q)
def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return ((a[0] * d) % Q, (-a[1] * d) % Q)
def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1: r = f2_mul(r, a)
        a = f2_mul(a, a); e >>= 1
    return r
F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (9, 1)
TWIST_B = f2_mul((3, 0), f2_inv(XI))   # G2: y^2 = x^3 + 3/(9+u)


# ---------------------------------------------------------------- generic affine curve ops (None = infinity)
class _F1:
    zero = 0
    @staticmethod
    def add(a, b): return (a + b) % Q
    @staticmethod
    def sub(a, b): return (a - b) % Q
    @staticmethod
    def mul(a, b): return (a * b) % Q
    @staticmethod
    def inv(a): return pow(a, -1, Q)
    @staticmethod
    def neg(a): return (-a) % Q
    @staticmethod
    def small(a, k): return (a * k) % Q

class _F2:
    zero = F2_ZERO
    add, sub, mul, inv, neg = staticmethod(f2_add), staticmethod(f2_sub), staticmethod(f2_mul), staticmethod(f2_inv), staticmethod(f2_neg)
    @staticmethod
    def small(a, k): return f2_muls(a, k)

def _pt_add(F, p, q):
    if p is None: return q
    if q is None: return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if y1 != y2 or y1 == F.zero:
            return None
        lam = F.mul(F.small(F.mul(x1, x1), 3), F.inv(F.small(y1, 2)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)

def _pt_mul(F, p, k):
    k %= R
    acc = None
    while k:
        if k & 1: acc = _pt_add(F, acc, p)
        p = _pt_add(F, p, p); k >>= 1
    return acc

def g1_add(p, q): return _pt_add(_F1, p, q)
def g1_mul(p, k): return _pt_mul(_F1, p, k)
def g1_neg(p): return None if p is None else (p[0], (-p[1]) % Q)
def g2_add(p, q): return _pt_add(_F2, p, q)
def g2_mul(p, k): return _pt_mul(_F2, p, k)
def g1_on_curve(p): return p is None or (p[1] * p[1] - p[0] ** 3 - 3) % Q == 0
def g2_on_curve(p):
    if p is None: return True
    x, y = p
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), TWIST_B)) == F2_ZERO


# ---------------------------------------------------------------- Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), lists of 12
def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):           # w^12 = 18 w^6 - 82
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [v % Q for v in t[:12]]

F12_ONE = [1] + [0] * 11

def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1: r = f12_mul(r, a)
        a = f12_mul(a, a); e >>= 1
    return r

def _embed(coeffs):
    """sum of (Fq2 a) * w^k terms -> Fq12 list, using u = w^6 - 9."""
    out = [0] * 12
    for k, a in coeffs:
        out[k] = (out[k] + a[0] - 9 * a[1]) % Q
        out[k + 6] = (out[k + 6] + a[1]) % Q
    return out

def _line(Rp, Tp, P):
    """Line through untwisted Rp, Tp (affine on the twist, Fq2) evaluated at P in G1; returns (l, Rp+Tp)."""
    (x1, y1), (x2, y2) = Rp, Tp
    if x1 == x2 and y1 == y2:
        lam = f2_mul(f2_muls(f2_mul(x1, x1), 3), f2_inv(f2_muls(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    xp, yp = P
    l = _embed([(0, (yp, 0)), (1, f2_neg(f2_muls(lam, xp))), (3, f2_sub(f2_mul(lam, x1), y1))])
    return l, (x3, y3)

def miller_loop(Qp, P):
    """Optimal ate Miller loop f_{6z+2,Q}(P) * l_{.,pi(Q)} * l_{.,-pi^2(Q)}; Qp in G2 (twist coords), P in G1."""
    if Qp is None or P is None:
        return F12_ONE
    f = F12_ONE
    Rp = Qp
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        l, Rp = _line(Rp, Rp, P)
        f = f12_mul(f12_mul(f, f), l)
        if (ATE_LOOP >> i) & 1:
            l, Rp = _line(Rp, Qp, P)
            f = f12_mul(f, l)
    g12 = f2_pow(XI, (Q - 1) // 3)
    g13 = f2_pow(XI, (Q - 1) // 2)
    Q1 = (f2_mul(f2_conj(Qp[0]), g12), f2_mul(f2_conj(Qp[1]), g13))
    g22 = f2_pow(XI, (Q * Q - 1) // 3)
    nQ2 = (f2_mul(Qp[0], g22), Qp[1])         # -pi^2(Q): y * xi^((p^2-1)/2) = -y, negated again
    l, Rp = _line(Rp, Q1, P)
    f = f12_mul(f, l)
    l, Rp = _line(Rp, nQ2, P)
    f = f12_mul(f, l)
    return f

def final_exp(f):
    return f12_pow(f, (Q ** 12 - 1) // R)

def pairing_product_is_one(pairs):
    f = F12_ONE
    for (p1, q2) in pairs:
        f = f12_mul(f, miller_loop(q2, p1))
    return final_exp(f) == F12_ONE


# ---------------------------------------------------------------- JSON <-> points  (src/export.cpp:56-121, src/import.cpp)
def _hx(v): return "0x%x" % v          # mpz_get_str(.,16,.): lowercase, no zero padding, zero -> "0"
def g1_from_json(a): return (int(a[0], 16), int(a[1], 16))
def g2_from_json(a):  # [[x.c1, x.c0], [y.c1, y.c0]]
    return ((int(a[0][1], 16), int(a[0][0], 16)), (int(a[1][1], 16), int(a[1][0], 16)))

def proof_to_json(A, B, C, inputs):
    """Exact text of src/export.cpp:99-121 (ALT_BN128 branch of :56-96)."""
    def g1(p):
        p = p if p is not None else (0, 1)   # libff affine form of zero is (0,1,0)
        return '"%s", "%s"' % (_hx(p[0]), _hx(p[1]))
    def g2(p):
        p = p if p is not None else ((0, 0), (1, 0))
        return '["%s", "%s"],\n ["%s", "%s"]' % (_hx(p[0][1]), _hx(p[0][0]), _hx(p[1][1]), _hx(p[1][0]))
    s = "{\n"
    s += ' "A" :[' + g1(A) + "],\n"
    s += ' "B"  :[' + g2(B) + "],\n"
    s += ' "C"  :[' + g1(C) + "],\n"
    s += ' "input" :[' + ", ".join('"%s"' % _hx(v) for v in inputs) + "]\n"
    s += "}"
    return s


# ---------------------------------------------------------------- verifier (tcc:552-670; ethsnarks/verifier.py:185-196)
def verify(vk, proof):
    """vk / proof are dicts in the reference's JSON schema (src/export.cpp:124-145, :99-121)."""
    alpha = g1_from_json(vk["alpha"]); beta = g2_from_json(vk["beta"])
    gamma = g2_from_json(vk["gamma"]); delta = g2_from_json(vk["delta"])
    ic = [g1_from_json(p) for p in vk["gammaABC"]]
    A = g1_from_json(proof["A"]); B = g2_from_json(proof["B"]); C = g1_from_json(proof["C"])
    inputs = [int(v, 16) for v in proof["input"]]
    if len(inputs) != len(ic) - 1:          # strong input consistency, tcc:646-654
        return False
    if not (g1_on_curve(A) and g2_on_curve(B) and g1_on_curve(C)):
        return False
    if g2_mul(B, R - 1) != (B[0], f2_neg(B[1])):   # subgroup check on G2 (cofactor != 1)
        return False
    acc = ic[0]
    for v, p in zip(inputs, ic[1:]):
        acc = g1_add(acc, g1_mul(p, v))
    return pairing_product_is_one([(A, B), (g1_neg(alpha), beta), (g1_neg(acc), gamma), (g1_neg(C), delta)])


# ---------------------------------------------------------------- evaluation domain (Appendix A.2)
def omega(m):
    k = m.bit_length() - 1
    assert 1 << k == m and k <= 28
    return pow(ROOT_2_28, 1 << (28 - k), R)

def ntt(a, w):
    n = len(a)
    if n == 1: return a[:]
    e = ntt(a[0::2], w * w % R); o = ntt(a[1::2], w * w % R)
    out = [0] * n; t = 1
    for i in range(n // 2):
        v = t * o[i] % R
        out[i] = (e[i] + v) % R; out[i + n // 2] = (e[i] - v) % R
        t = t * w % R
    return out

def intt(a):
    m = len(a); mi = pow(m, -1, R)
    return [v * mi % R for v in ntt(a, pow(omega(m), -1, R))]

def coset_ntt(a, g):
    m = len(a)
    return ntt([v * pow(g, i, R) % R for i, v in enumerate(a)], omega(m))

def coset_intt(a, g):
    gi = pow(g, -1, R)
    return [v * pow(gi, i, R) % R for i, v in enumerate(intt(a))]


# ---------------------------------------------------------------- R1CS helpers
# An R1CS here is (nC, nIn, V, A, B, C) with A/B/C lists of rows, each row a list of (var_index, coeff),
# var_index 0 = constant ONE.  Witness w has V+1 entries, w[0] = 1.
def domain_size(nC, nIn):
    v = nC + nIn + 1                      # src/stubs.cpp:65
    m = 1
    while m < v: m <<= 1                  # roundUpToNearestPowerOf2, src/stubs.cpp:49-59
    return m

def witness_map(r1cs, w):
    """r1cs_to_qap_witness_map with d1=d2=d3=0 (Appendix A.3). Returns h[0..m] (m+1 entries)."""
    nC, nIn, V, A, B, C = r1cs
    m = domain_size(nC, nIn)
    dot = lambda row: sum(c * w[i] for i, c in row) % R
    aA = [dot(r) for r in A] + [w[i] for i in range(nIn + 1)]
    aA += [0] * (m - len(aA))
    aB = [dot(r) for r in B] + [0] * (m - nC)
    aC = [dot(r) for r in C] + [0] * (m - nC)
    aA = coset_ntt(intt(aA), COSET_G); aB = coset_ntt(intt(aB), COSET_G); aC = coset_ntt(intt(aC), COSET_G)
    zinv = pow(pow(COSET_G, m, R) - 1, -1, R)
    H = [((a * b - c) * zinv) % R for a, b, c in zip(aA, aB, aC)]
    return coset_intt(H, COSET_G) + [0]


def prove(pk, r1cs, w):
    """pk: dict(alpha_g1, beta_g2, A=(indices, points), B=(indices, points), H=[...], L=[...]) affine ints.
    Returns (A, B, C) affine.  Restates tcc:451-550."""
    nC, nIn, V, _, _, _ = r1cs
    m = domain_size(nC, nIn)
    h = witness_map(r1cs, w)
    assert h[m - 1] == 0 and h[m] == 0
    At = None
    for i, p in zip(*pk["A"]): At = g1_add(At, g1_mul(p, w[i]))
    Bt = None
    for i, p in zip(*pk["B"]): Bt = g2_add(Bt, g2_mul(p, w[i]))
    Ht = None
    for j in range(m - 1): Ht = g1_add(Ht, g1_mul(pk["H"][j], h[j]))
    Lt = None
    for k, p in enumerate(pk["L"]): Lt = g1_add(Lt, g1_mul(p, w[nIn + 1 + k]))
    return g1_add(pk["alpha_g1"], At), g2_add(pk["beta_g2"], Bt), g1_add(Ht, Lt)


def proof_from_trapdoor(r1cs, w, t, alpha, beta, gamma, delta, g1=G1_GEN, g2=G2_GEN):
    """The no-ZK proof in closed form from the toxic waste, straight from the definitions the reference's comments give
    (tcc:533-540 with the key elements of tcc:326-350 and the QAP of SURVEY Appendix A.3/A.4):
        A = (alpha + sum_i w_i A_i(t)) G1,   B = (beta + sum_i w_i B_i(t)) G2,
        C = ((sum_{i>nIn} w_i (beta A_i + alpha B_i + C_i)(t) + A(t) B(t) - C(t)) / delta) G1,
    where A(t) = sum_i w_i A_i(t) etc. and H(t) Z(t) = A(t) B(t) - C(t).  Three scalar multiplications, O(nnz) field
    operations: no multi-exponentiation, no transform, no key, no codec -- an independent pin of prove() / the C oracle /
    the HIP path for any (r1cs, witness, toxic waste).  (gamma only enters the verification key.)"""
    nC, nIn, V, A, B, C = r1cs
    m = domain_size(nC, nIn)
    om = omega(m)
    Zt = (pow(t, m, R) - 1) % R
    # Lagrange basis at t, u_j = omega^j Z(t) / (m (t - omega^j)): one batch inversion
    wj, den, pre, acc = 1, [], [], 1
    for j in range(m):
        d = m * (t - wj) % R
        den.append(d); pre.append(acc); acc = acc * d % R
        wj = wj * om % R
    inv = pow(acc, -1, R)
    u = [0] * m
    wpow = pow(om, m - 1, R); omi = pow(om, -1, R)
    for j in range(m - 1, -1, -1):
        u[j] = wpow * Zt % R * (inv * pre[j] % R) % R
        inv = inv * den[j] % R
        wpow = wpow * omi % R
    def evaluate(rows):
        """(sum over all variables, sum over the public ones i <= nIn) of w_i M_i(t), M_i(t) = sum_j M_ji u_j"""
        tot = pub = 0
        for j, row in enumerate(rows):
            for i, c in row:
                v = c * w[i] % R * u[j]
                tot += v
                if i <= nIn: pub += v
        return tot % R, pub % R
    SA, pA = evaluate(A); SB, pB = evaluate(B); SC, pC = evaluate(C)
    for i in range(nIn + 1):                                   # input-consistency rows of A (Appendix A.3 step 1)
        v = u[nC + i] * w[i] % R
        SA = (SA + v) % R; pA = (pA + v) % R
    priv = (beta * (SA - pA) + alpha * (SB - pB) + (SC - pC)) % R
    c = (priv + SA * SB - SC) * pow(delta, -1, R) % R
    return g1_mul(g1, (alpha + SA) % R), g2_mul(g2, (beta + SB) % R), g1_mul(g1, c)


def keygen(r1cs, t, alpha, beta, gamma, delta, g1=G1_GEN, g2=G2_GEN):
    """Generator tcc:277-449 + Appendix A.4 + nozk conversion hpp:209-233, toxic waste given explicitly."""
    nC, nIn, V, A, B, C = r1cs
    m = domain_size(nC, nIn)
    w = omega(m)
    Zt = (pow(t, m, R) - 1) % R
    mi = pow(m, -1, R)
    u = [(pow(w, i, R) * Zt % R) * pow(m * (t - pow(w, i, R)) % R, -1, R) % R for i in range(m)]
    At = [0] * (V + 1); Bt = [0] * (V + 1); Ct = [0] * (V + 1)
    for i in range(nIn + 1): At[i] = u[nC + i]
    for j in range(nC):
        for i, c in A[j]: At[i] = (At[i] + u[j] * c) % R
        for i, c in B[j]: Bt[i] = (Bt[i] + u[j] * c) % R
        for i, c in C[j]: Ct[i] = (Ct[i] + u[j] * c) % R
    gi = pow(gamma, -1, R); di = pow(delta, -1, R)
    abc = [(beta * At[i] + alpha * Bt[i] + Ct[i]) % R for i in range(V + 1)]
    gamma_abc = [abc[i] * gi % R for i in range(nIn + 1)]
    Ls = [abc[i] * di % R for i in range(nIn + 1, V + 1)]
    Hs = [pow(t, j, R) * Zt % R * di % R for j in range(m - 1)]
    pk = dict(
        alpha_g1=g1_mul(g1, alpha), beta_g1=g1_mul(g1, beta), beta_g2=g2_mul(g2, beta),
        delta_g1=g1_mul(g1, delta), delta_g2=g2_mul(g2, delta),
        A=([i for i in range(V + 1) if At[i]], [g1_mul(g1, At[i]) for i in range(V + 1) if At[i]]),
        B=([i for i in range(V + 1) if Bt[i]], [g2_mul(g2, Bt[i]) for i in range(V + 1) if Bt[i]]),
        H=[g1_mul(g1, s) for s in Hs], L=[g1_mul(g1, s) for s in Ls])
    vk = dict(alpha_g1=pk["alpha_g1"], beta_g2=pk["beta_g2"], gamma_g2=g2_mul(g2, gamma),
              delta_g2=pk["delta_g2"], gamma_abc=[g1_mul(g1, s) for s in gamma_abc])
    return pk, vk


def vk_to_json_dict(vk):
    g1 = lambda p: [_hx(p[0]), _hx(p[1])]
    g2 = lambda p: [[_hx(p[0][1]), _hx(p[0][0])], [_hx(p[1][1]), _hx(p[1][0])]]
    return {"alpha": g1(vk["alpha_g1"]), "beta": g2(vk["beta_g2"]), "gamma": g2(vk["gamma_g2"]),
            "delta": g2(vk["delta_g2"]), "gammaABC": [g1(p) for p in vk["gamma_abc"]]}


# ------------------------------------------------------------------ bellman-style proving-key JSON
def _jac_g1(a):
    """readG1 (src/export.cpp:223-236): Jacobian decimal triple -> affine (None = zero)"""
    x, y, z = (int(v) % Q for v in a)
    if z == 0: return None
    zi = pow(z, -1, Q)
    return (x * zi * zi % Q, y * zi * zi % Q * zi % Q)


def _jac_g2(a):
    """readG2 (src/export.cpp:238-265): [[x.c0, x.c1], [y.c0, y.c1], [z.c0, z.c1]] -> affine"""
    x, y, z = ((int(v[0]) % Q, int(v[1]) % Q) for v in a)
    if z == (0, 0): return None
    zi = f2_inv(z); zi2 = f2_mul(zi, zi)
    return (f2_mul(x, zi2), f2_mul(y, f2_mul(zi2, zi)))


def pk_from_bellman_json(d):
    """pk_bellman2ethsnarks (src/export.cpp:267-321) + the nozk conversion (hpp:209-233) on a parsed JSON dict:
    A keeps its non-zero entries, B keeps B2[i] where B1[i] is non-zero (domain |A|), L = C[2..], H = hExps.
    Returns the pk dict of keygen() plus the two domain sizes; points are affine or None."""
    A = [_jac_g1(p) for p in d["A"]]
    B1 = [_jac_g1(p) for p in d["B1"]]
    B2 = [_jac_g2(p) for p in d["B2"]]
    keepA = [i for i, p in enumerate(A) if p is not None]
    keepB = [i for i, p in enumerate(B1) if p is not None]
    return dict(alpha_g1=_jac_g1(d["vk_alfa_1"]), beta_g1=_jac_g1(d["vk_beta_1"]), beta_g2=_jac_g2(d["vk_beta_2"]),
                delta_g1=_jac_g1(d["vk_delta_1"]), delta_g2=_jac_g2(d["vk_delta_2"]),
                A=(keepA, [A[i] for i in keepA]), B=(keepB, [B2[i] for i in keepB]),
                H=[_jac_g1(p) for p in d["hExps"]], L=[_jac_g1(p) for p in d["C"][2:]],
                a_domain=len(A), b_domain=len(A))
