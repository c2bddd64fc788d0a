// verify.cpp -- Groth16 verifier, host C++ like the reference's (the reference verifies on the
// CPU too: r1cs_gg_ppzksnark_zok_verifier_strong_IC, r1cs_gg_ppzksnark_zok.tcc:552-670, called from
// stub_verify src/stubs.cpp:16-33 and exported as `bool ethsnarks_verify(vk_json, proof_json)` by
// src/verify_dll.cpp:3-10).  JSON in: vk2json / proof_to_json text (src/export.cpp:99-145, parsed like
// src/import.cpp:34-223: hex or decimal strings, Fq2 as [c1, c0]).
//
// Equation (tcc:597-610; ethsnarks/verifier.py:185-196):  e(A,B) = e(alpha,beta) e(acc,gamma) e(C,delta)
// with acc = gammaABC[0] + sum input_i gammaABC[i+1], strong input consistency |input| = |gammaABC| - 1.
// Pairing: optimal ate on BN254, computed the plain way -- Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), affine
// Miller loop over the twist, final exponentiation by the full (q^12 - 1)/r -- because verification is
// off the proving path; it is here to close the keygen -> prove -> verify loop without py_ecc or an EVM.
#include <memory>
#include <string>
#include <vector>
#include <string.h>
#include "bn254.hpp"
#include "keygen.hpp"
#include "../../include/zkhip.h"

using namespace zk;

namespace {
int vfail(int code, const char *msg) { snprintf(g_last_error, sizeof(g_last_error), "%s", msg); return code; }

// ---------------------------------------------------------------- Fq12, polynomial basis
struct F12 { fe c[12]; };
const fe K18 = Fq::from_u64(18), K82 = Fq::from_u64(82), K9 = Fq::from_u64(9);
F12 f12_one() { F12 r; for (int i = 0; i < 12; i++) r.c[i] = Fq::zero(); r.c[0] = Fq::one(); return r; }
bool f12_is_one(const F12 &a) {
    if (!Fq::eq(a.c[0], Fq::one())) return false;
    for (int i = 1; i < 12; i++) if (!Fq::is_zero(a.c[i])) return false;
    return true;
}
F12 f12_mul(const F12 &a, const F12 &b) {
    fe t[23];
    for (int i = 0; i < 23; i++) t[i] = Fq::zero();
    for (int i = 0; i < 12; i++) {
        if (Fq::is_zero(a.c[i])) continue;
        for (int j = 0; j < 12; j++) {
            if (Fq::is_zero(b.c[j])) continue;
            t[i + j] = Fq::add(t[i + j], Fq::mul(a.c[i], b.c[j]));
        }
    }
    for (int k = 22; k >= 12; k--) {                  // w^12 = 18 w^6 - 82
        if (Fq::is_zero(t[k])) continue;
        t[k - 6] = Fq::add(t[k - 6], Fq::mul(t[k], K18));
        t[k - 12] = Fq::sub(t[k - 12], Fq::mul(t[k], K82));
    }
    F12 r;
    for (int i = 0; i < 12; i++) r.c[i] = t[i];
    return r;
}
// (Fq2 a) * w^k embedded with u = w^6 - 9
void f12_add_term(F12 &r, int k, const fe2 &a) {
    r.c[k] = Fq::add(r.c[k], Fq::sub(a.c0, Fq::mul(a.c1, K9)));
    r.c[k + 6] = Fq::add(r.c[k + 6], a.c1);
}
const uint32_t FINAL_EXP[88] = {0xca86f120u, 0x86964b64u, 0xe54523a4u, 0x40a4efb7u, 0x96e84abbu, 0x837fa978u, 0xb9b2b918u, 0x361102b6u, 0xf35692dau, 0xc0de81deu, 0xa6c3c760u, 0xbe04c7e8u, 0xd570bb7fu, 0xd766f9c9u, 0x83561841u, 0xc230974du, 0xc3be69a3u, 0x5bba1668u, 0x10526294u, 0x7f3811c4u, 0xdadda71cu, 0x29baee7du, 0x145da900u, 0xbf813b8du, 0x423f9a2cu, 0x641bbadfu, 0x44eacc5eu, 0xa80bb4eau, 0x14fde37cu, 0xcd656648u, 0x580291d2u, 0x4a0364b9u, 0x0826f0ddu, 0xee93dfb1u, 0xc5514724u, 0x6b42db8du, 0x0b0f3785u, 0xbb10cf43u, 0x6f804216u, 0x40494e40u, 0xacf3aafbu, 0x55cfe107u, 0xe0ebae87u, 0x2088ec80u, 0x11a337a0u, 0x846a3ed0u, 0x1e3a5195u, 0x48a45a4au, 0xdfc50e16u, 0xe5664568u, 0x4c0cc4ebu, 0xab6a4129u, 0xd268c7dau, 0x82d0d602u, 0xed3cc48au, 0x6668449au, 0xb2015dfcu, 0x5062cd0fu, 0xb1ddb3d1u, 0x7f2940a8u, 0x2a226448u, 0x77f5b63au, 0x61e443aeu, 0xfef07813u, 0x88d5c6c8u, 0xf977870eu, 0x1f676baau, 0x790364a6u, 0xceaddea3u, 0x5887e72eu, 0xa09a1b70u, 0x1377e563u, 0x1bd8c3b2u, 0x0c54efeeu, 0xd524d8f7u, 0x3ec3d15au, 0xb2383a5du, 0xdaf15466u, 0xbb94fec0u, 0xe1e30a73u, 0x5f3f7be2u, 0x6a1c7101u, 0x6369b1ffu, 0x842d43bfu, 0x107d20bcu, 0x20fddadfu, 0x4b6dc970u, 0x0000002fu};
F12 final_exp(const F12 &f) {
    F12 acc = f12_one(), base = f;
    for (int i = 0; i < 88 * 32; i++) {
        if ((FINAL_EXP[i >> 5] >> (i & 31)) & 1) acc = f12_mul(acc, base);
        base = f12_mul(base, base);
    }
    return acc;
}

fe2 f2_mul_fq(const fe2 &a, const fe &s) { fe2 r; r.c0 = Fq::mul(a.c0, s); r.c1 = Fq::mul(a.c1, s); return r; }
fe2 f2_conj(const fe2 &a) { fe2 r; r.c0 = a.c0; r.c1 = Fq::neg(a.c1); return r; }
fe2 f2_pow(const fe2 &a, const uint32_t *e, int nwords) {
    fe2 acc = Fq2::one(), base = a;
    for (int i = 0; i < nwords * 32; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = Fq2::mul(acc, base);
        base = Fq2::sqr(base);
    }
    return acc;
}

struct P2 { fe2 x, y; };
// line through R, T on the twist evaluated at P in G1 (see oracle/pyref.py _line); R <- R + T
F12 line_and_add(P2 &R, const P2 &T, const G1::Affine &P) {
    fe2 lam;
    if (Fq2::eq(R.x, T.x) && Fq2::eq(R.y, T.y)) {
        fe2 xx = Fq2::sqr(R.x);
        lam = Fq2::mul(Fq2::add(Fq2::dbl(xx), xx), Fq2::inv(Fq2::dbl(R.y)));
    } else {
        lam = Fq2::mul(Fq2::sub(T.y, R.y), Fq2::inv(Fq2::sub(T.x, R.x)));
    }
    fe2 x3 = Fq2::sub(Fq2::sub(Fq2::sqr(lam), R.x), T.x);
    fe2 y3 = Fq2::sub(Fq2::mul(lam, Fq2::sub(R.x, x3)), R.y);
    F12 l; for (int i = 0; i < 12; i++) l.c[i] = Fq::zero();
    l.c[0] = P.y;                                                        // y_P
    f12_add_term(l, 1, Fq2::neg(f2_mul_fq(lam, P.x)));                   // - lam x_P w
    f12_add_term(l, 3, Fq2::sub(Fq2::mul(lam, R.x), R.y));              // (lam x_R - y_R) w^3
    R.x = x3; R.y = y3;
    return l;
}

// q as 8 x u32 and the exponents (q-1)/3, (q-1)/2, (q^2-1)/3
void big_from_fe_params(std::vector<uint32_t> &q) { q.resize(8); for (int i = 0; i < 8; i++) q[i] = FqParams::p(i); }
std::vector<uint32_t> big_mul(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b) {
    std::vector<uint32_t> r(a.size() + b.size(), 0);
    for (size_t i = 0; i < a.size(); i++) {
        uint64_t c = 0;
        for (size_t j = 0; j < b.size(); j++) { c += (uint64_t)a[i] * b[j] + r[i + j]; r[i + j] = (uint32_t)c; c >>= 32; }
        r[i + b.size()] = (uint32_t)c;
    }
    return r;
}
void big_sub1(std::vector<uint32_t> &a) { for (size_t i = 0; i < a.size(); i++) { if (a[i]--) break; } }
void big_div_small(std::vector<uint32_t> &a, uint32_t d) {
    uint64_t rem = 0;
    for (size_t i = a.size(); i-- > 0;) { uint64_t cur = (rem << 32) | a[i]; a[i] = (uint32_t)(cur / d); rem = cur % d; }
}

F12 miller_loop(const G2::Affine &Qa, const G1::Affine &P) {
    if (G2::is_inf(Qa) || G1::is_inf(P)) return f12_one();
    const uint64_t ATE = 0x9d797039be763ba8ull;                         // low 64 bits of 6z+2 = 29793968203157093288 (bit 64 is the implicit top bit)
    P2 Q{Qa.x, Qa.y}, R = Q;
    F12 f = f12_one();
    for (int i = 63; i >= 0; i--) {
        P2 Rc = R;
        F12 l = line_and_add(R, Rc, P);
        f = f12_mul(f12_mul(f, f), l);
        if ((ATE >> i) & 1) { l = line_and_add(R, Q, P); f = f12_mul(f, l); }
    }
    std::vector<uint32_t> q, e;
    big_from_fe_params(q);
    fe2 xi; xi.c0 = K9; xi.c1 = Fq::one();
    e = q; big_sub1(e); big_div_small(e, 3);
    fe2 g12 = f2_pow(xi, e.data(), (int)e.size());
    e = q; big_sub1(e); big_div_small(e, 2);
    fe2 g13 = f2_pow(xi, e.data(), (int)e.size());
    e = big_mul(q, q); big_sub1(e); big_div_small(e, 3);
    fe2 g22 = f2_pow(xi, e.data(), (int)e.size());
    P2 Q1{Fq2::mul(f2_conj(Q.x), g12), Fq2::mul(f2_conj(Q.y), g13)};
    P2 nQ2{Fq2::mul(Q.x, g22), Q.y};                                    // -pi^2(Q)
    F12 l = line_and_add(R, Q1, P); f = f12_mul(f, l);
    l = line_and_add(R, nQ2, P); f = f12_mul(f, l);
    return f;
}

// ---------------------------------------------------------------- JSON (src/import.cpp:34-223)
// value of `key`: the hex/decimal strings inside its bracket-balanced array, in order
bool json_strings(const std::string &js, const char *key, std::vector<std::string> &out) {
    std::string pat = std::string("\"") + key + "\"";
    size_t p = js.find(pat);
    if (p == std::string::npos) return false;
    p = js.find(':', p + pat.size());
    if (p == std::string::npos) return false;
    p = js.find('[', p);
    if (p == std::string::npos) return false;
    int depth = 0;
    for (; p < js.size(); p++) {
        char ch = js[p];
        if (ch == '[') depth++;
        else if (ch == ']') { if (--depth == 0) return true; }
        else if (ch == '"') {
            size_t q = js.find('"', p + 1);
            if (q == std::string::npos) return false;
            out.push_back(js.substr(p + 1, q - p - 1));
            p = q;
        }
    }
    return false;
}
// parse_bigint (src/import.hpp:15-33): "0x" hex or decimal -> canonical limbs; must be < 2^256
bool parse_big(const std::string &s, fe &out) {
    uint32_t l[9] = {0};
    size_t i = 0; uint32_t base = 10;
    if (s.size() > 2 && s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) { base = 16; i = 2; }
    if (i >= s.size()) return false;
    for (; i < s.size(); i++) {
        char ch = s[i]; uint32_t d;
        if (ch >= '0' && ch <= '9') d = ch - '0';
        else if (base == 16 && ch >= 'a' && ch <= 'f') d = ch - 'a' + 10;
        else if (base == 16 && ch >= 'A' && ch <= 'F') d = ch - 'A' + 10;
        else return false;
        uint64_t c = d;
        for (int k = 0; k < 9; k++) { c += (uint64_t)l[k] * base; l[k] = (uint32_t)c; c >>= 32; }
        if (l[8]) return false;
    }
    for (int k = 0; k < 8; k++) out.l[k] = l[k];
    return true;
}
bool lt_modulus_q(const fe &a) { for (int i = 7; i >= 0; i--) { if (a.l[i] != FqParams::p(i)) return a.l[i] < FqParams::p(i); } return false; }
bool lt_modulus_r(const fe &a) { for (int i = 7; i >= 0; i--) { if (a.l[i] != FrParams::p(i)) return a.l[i] < FrParams::p(i); } return false; }
bool fq_from_str(const std::string &s, fe &out) { fe c; if (!parse_big(s, c) || !lt_modulus_q(c)) return false; out = Fq::to_mont(c); return true; }
bool g1_from(const std::vector<std::string> &v, size_t at, G1::Affine &p) { return at + 2 <= v.size() && fq_from_str(v[at], p.x) && fq_from_str(v[at + 1], p.y); }
bool g2_from(const std::vector<std::string> &v, size_t at, G2::Affine &p) {   // [[x.c1, x.c0], [y.c1, y.c0]]
    return at + 4 <= v.size() && fq_from_str(v[at], p.x.c1) && fq_from_str(v[at + 1], p.x.c0) && fq_from_str(v[at + 2], p.y.c1) && fq_from_str(v[at + 3], p.y.c0);
}
bool g1_on_curve(const G1::Affine &p) {
    if (G1::is_inf(p)) return true;
    return Fq::eq(Fq::sqr(p.y), Fq::add(Fq::mul(Fq::sqr(p.x), p.x), Fq::from_u64(3)));
}
fe2 twist_b() { fe2 xi; xi.c0 = K9; xi.c1 = Fq::one(); fe2 three; three.c0 = Fq::from_u64(3); three.c1 = Fq::zero(); return Fq2::mul(three, Fq2::inv(xi)); }
bool g2_on_curve(const G2::Affine &p) {
    if (G2::is_inf(p)) return true;
    return Fq2::eq(Fq2::sqr(p.y), Fq2::add(Fq2::mul(Fq2::sqr(p.x), p.x), twist_b()));
}
template <class C> typename C::XYZZ scalar_mul(const typename C::Affine &p, const fe &k_canon) {
    typename C::XYZZ acc = C::infinity();
    for (int i = 255; i >= 0; i--) {
        acc = C::dbl(acc);
        if ((k_canon.l[i >> 5] >> (i & 31)) & 1) acc = C::madd(acc, p);
    }
    return acc;
}

// vk_from_json (src/import.cpp:195-223): keys alpha, beta, gamma, delta, gammaABC
bool parse_vk(const std::string &vk, zk_vk &out) {
    std::vector<std::string> s;
    bool ok = true;
    s.clear(); ok = ok && json_strings(vk, "alpha", s) && s.size() == 2 && g1_from(s, 0, out.alpha_g1);
    s.clear(); ok = ok && json_strings(vk, "beta", s) && s.size() == 4 && g2_from(s, 0, out.beta_g2);
    s.clear(); ok = ok && json_strings(vk, "gamma", s) && s.size() == 4 && g2_from(s, 0, out.gamma_g2);
    s.clear(); ok = ok && json_strings(vk, "delta", s) && s.size() == 4 && g2_from(s, 0, out.delta_g2);
    s.clear(); ok = ok && json_strings(vk, "gammaABC", s) && s.size() >= 2 && s.size() % 2 == 0;
    if (ok) { out.gamma_abc.resize(s.size() / 2); for (size_t i = 0; i < out.gamma_abc.size() && ok; i++) ok = g1_from(s, 2 * i, out.gamma_abc[i]); }
    return ok;
}
// proof_from_json (src/import.cpp:161-192): keys A, B, C, input; inputs come back canonical
bool parse_proof(const std::string &pf, G1::Affine &A, G2::Affine &B, G1::Affine &Cc, std::vector<fe> &inputs) {
    std::vector<std::string> s;
    bool ok = true;
    s.clear(); ok = ok && json_strings(pf, "A", s) && s.size() == 2 && g1_from(s, 0, A);
    s.clear(); ok = ok && json_strings(pf, "B", s) && s.size() == 4 && g2_from(s, 0, B);
    s.clear(); ok = ok && json_strings(pf, "C", s) && s.size() == 2 && g1_from(s, 0, Cc);
    s.clear(); ok = ok && json_strings(pf, "input", s);
    if (ok) { inputs.resize(s.size()); for (size_t i = 0; i < s.size() && ok; i++) ok = parse_big(s[i], inputs[i]) && lt_modulus_r(inputs[i]); }
    return ok;
}
}  // namespace

extern "C" int zk_vk_from_json(const char *vk_json, zk_vk **out) try {
    if (!vk_json || !out) return vfail(ZK_ERR_ARG, "null argument");
    std::unique_ptr<zk_vk> vk(new zk_vk());
    if (!parse_vk(vk_json, *vk)) return vfail(ZK_ERR_FORMAT, "cannot parse verification key JSON");
    *out = vk.release();
    return ZK_OK;
} ZK_GUARD

extern "C" int zk_proof_from_json(const char *proof_json, zk_proof *out, uint64_t *inputs_canon, uint32_t cap, uint32_t *n_inputs) try {
    if (!proof_json || !out || !n_inputs) return vfail(ZK_ERR_ARG, "null argument");
    G1::Affine A, Cc; G2::Affine B; std::vector<fe> in;
    if (!parse_proof(proof_json, A, B, Cc, in)) return vfail(ZK_ERR_FORMAT, "cannot parse proof JSON");
    memset(out, 0, sizeof(*out));
    auto put = [](uint64_t dst[4], const fe &mont) { fe c = Fq::from_mont(mont); memcpy(dst, c.l, 32); };
    put(out->a_x, A.x); put(out->a_y, A.y);
    put(out->b_x_c0, B.x.c0); put(out->b_x_c1, B.x.c1); put(out->b_y_c0, B.y.c0); put(out->b_y_c1, B.y.c1);
    put(out->c_x, Cc.x); put(out->c_y, Cc.y);
    *n_inputs = (uint32_t)in.size();
    if (in.size() > cap || (in.size() && !inputs_canon)) return vfail(ZK_ERR_BUFFER, "input buffer too small");
    for (size_t i = 0; i < in.size(); i++) memcpy(inputs_canon + 4 * i, in[i].l, 32);
    return ZK_OK;
} ZK_GUARD

extern "C" int zk_verify(const char *vk_json, const char *proof_json, int *accepted) try {
    if (!vk_json || !proof_json || !accepted) return vfail(ZK_ERR_ARG, "null argument");
    *accepted = 0;
    zk_vk key;
    G1::Affine A, Cc; G2::Affine B;
    std::vector<fe> inputs;
    if (!parse_vk(vk_json, key) || !parse_proof(proof_json, A, B, Cc, inputs))
        return vfail(ZK_ERR_FORMAT, "cannot parse verification key / proof JSON");   // reference: std::invalid_argument / json exception
    const G1::Affine &alpha = key.alpha_g1; const G2::Affine &beta = key.beta_g2, &gamma = key.gamma_g2, &delta = key.delta_g2;
    const std::vector<G1::Affine> &ic = key.gamma_abc;
    if (inputs.size() + 1 != ic.size()) return ZK_OK;                         // strong input consistency, tcc:646-654
    // well-formedness, tcc:585-592 (points on their curves; B in the order-r subgroup of the twist)
    if (!g1_on_curve(A) || !g1_on_curve(Cc) || !g2_on_curve(B)) return ZK_OK;
    for (auto &p : ic) if (!g1_on_curve(p)) return ZK_OK;
    if (!g1_on_curve(alpha) || !g2_on_curve(beta) || !g2_on_curve(gamma) || !g2_on_curve(delta)) return ZK_OK;
    { fe rm; for (int i = 0; i < 8; i++) rm.l[i] = FrParams::p(i); if (!G2::is_inf(scalar_mul<G2>(B, rm))) return ZK_OK; }
    G1::XYZZ acc = G1::from_affine(ic[0]);                                    // tcc:578
    for (size_t i = 0; i < inputs.size(); i++) acc = G1::add(acc, scalar_mul<G1>(ic[i + 1], inputs[i]));
    const G1::Affine accA = G1::to_affine(acc);
    // e(A,B) e(-alpha,beta) e(-acc,gamma) e(-C,delta) == 1
    F12 f = miller_loop(B, A);
    f = f12_mul(f, miller_loop(beta, G1::neg(alpha)));
    f = f12_mul(f, miller_loop(gamma, G1::neg(accA)));
    f = f12_mul(f, miller_loop(delta, G1::neg(Cc)));
    *accepted = f12_is_one(final_exp(f)) ? 1 : 0;
    return ZK_OK;
} ZK_GUARD

// drop-in for libethsnarks_verify (src/verify_dll.cpp:3-10): true iff the proof verifies
extern "C" bool ethsnarks_verify(const char *vk_json, const char *proof_json) try {
    int ok = 0;
    return zk_verify(vk_json, proof_json, &ok) == ZK_OK && ok == 1;
} ZK_GUARD_BOOL
