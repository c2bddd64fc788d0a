// pkjson.cpp -- the reference's second proving-key source: a bellman / snarkjs style JSON
//   pk_bellman2ethsnarks(bellman_pk_file, pk_file)   src/export.cpp:267-328
//   readG1 / readG2                                  src/export.cpp:223-265
// followed by the nozk conversion of r1cs_gg_ppzksnark_zok.hpp:209-233.  Host-only (a file-format codec, no field
// arithmetic beyond the projective -> affine normalisation the `.raw` writer performs anyway).
//
// JSON keys: "A", "B1", "B2", "C", "hExps" (arrays of points), "vk_alfa_1", "vk_beta_1", "vk_delta_1" (G1),
// "vk_beta_2", "vk_delta_2" (G2).  A G1 point is ["x", "y", "z"], a G2 point is [["x.c0","x.c1"], ["y.c0","y.c1"],
// ["z.c0","z.c1"]] -- decimal strings, Jacobian coordinates (libff alt_bn128_G1(x, y, z)), z = 0 is the point at
// infinity.  Mapping (export.cpp:281-321):
//   A_query  = A, zero entries dropped, indices kept            (hpp:216-224)
//   B_query  = B2[i] for every i whose B1[i] is not zero          (export.cpp:290-301; domain = |A|)
//   L_query  = C[2..]  (the reference hard-codes one public input, export.cpp:305)
//   H_query  = hExps
#include <string>
#include <vector>
#include <stdio.h>
#include <string.h>
#include "bn254.hpp"
#include "../../include/zkhip.h"

using namespace zk;

namespace {
int jfail(int code, const std::string &msg) { snprintf(g_last_error, sizeof(g_last_error), "%s", msg.c_str()); return code; }

// decimal string -> Montgomery Fq; any decimal is accepted and reduced mod q (libff's Fp_model(const char*) takes
// numerals below the modulus; larger ones are undefined there)
bool fq_from_decimal(const char *s, size_t len, fe &out) {
    if (!len) return false;
    fe acc = Fq::zero();
    const fe ten9 = Fq::from_u64(1000000000u);
    size_t i = 0;
    while (i < len) {                                   // 9 digits at a time: acc = acc * 10^k + chunk
        size_t k = (len - i) % 9; if (!k) k = 9;
        uint64_t chunk = 0, scale = 1;
        for (size_t j = 0; j < k; j++) { char ch = s[i + j]; if (ch < '0' || ch > '9') return false; chunk = chunk * 10 + (uint64_t)(ch - '0'); scale *= 10; }
        acc = Fq::add(Fq::mul(acc, k == 9 ? ten9 : Fq::from_u64(scale)), Fq::from_u64(chunk));
        i += k;
    }
    out = acc;
    return true;
}

// the quoted strings inside the bracket-balanced array that is the value of top-level key `key`
struct Span { const char *p; size_t n; };
bool key_strings(const std::string &js, const char *key, std::vector<Span> &out) {
    const std::string pat = std::string("\"") + key + "\"";
    size_t p = 0;
    for (;;) {                                          // the key itself, not a value that happens to spell it
        p = js.find(pat, p);
        if (p == std::string::npos) return false;
        size_t q = p + pat.size();
        while (q < js.size() && (js[q] == ' ' || js[q] == '\t' || js[q] == '\n' || js[q] == '\r')) q++;
        if (q < js.size() && js[q] == ':') { p = q + 1; break; }
        p += pat.size();
    }
    p = js.find('[', p);
    if (p == std::string::npos) return false;
    int depth = 0;
    for (; p < js.size(); p++) {
        const char ch = js[p];
        if (ch == '[') depth++;
        else if (ch == ']') { if (--depth == 0) return true; }
        else if (ch == '"') {
            size_t q = js.find('"', p + 1);
            if (q == std::string::npos) return false;
            out.push_back(Span{js.data() + p + 1, q - p - 1});
            p = q;
        } else if (ch == '{' || ch == '}') return false;
    }
    return false;
}

struct JacG1 { fe x, y, z; };
struct JacG2 { fe2 x, y, z; };
bool read_g1s(const std::string &js, const char *key, std::vector<JacG1> &out) {
    std::vector<Span> s;
    if (!key_strings(js, key, s) || s.size() % 3) return false;
    out.resize(s.size() / 3);
    for (size_t i = 0; i < out.size(); i++)
        if (!fq_from_decimal(s[3 * i].p, s[3 * i].n, out[i].x) || !fq_from_decimal(s[3 * i + 1].p, s[3 * i + 1].n, out[i].y) ||
            !fq_from_decimal(s[3 * i + 2].p, s[3 * i + 2].n, out[i].z)) return false;
    return true;
}
bool read_g2s(const std::string &js, const char *key, std::vector<JacG2> &out) {
    std::vector<Span> s;
    if (!key_strings(js, key, s) || s.size() % 6) return false;
    out.resize(s.size() / 6);
    for (size_t i = 0; i < out.size(); i++) {
        fe *dst[6] = {&out[i].x.c0, &out[i].x.c1, &out[i].y.c0, &out[i].y.c1, &out[i].z.c0, &out[i].z.c1};   // Fq2(c0, c1), export.cpp:246-257
        for (int k = 0; k < 6; k++) if (!fq_from_decimal(s[6 * i + k].p, s[6 * i + k].n, *dst[k])) return false;
    }
    return true;
}

// Jacobian -> affine for a whole vector with one inversion (Montgomery's trick); z = 0 -> (0, 0)
template <class F, class E, class J, class A> void normalise(const std::vector<J> &in, std::vector<A> &out) {
    const size_t n = in.size();
    out.resize(n);
    std::vector<E> pre(n);
    E run = F::one();
    for (size_t i = 0; i < n; i++) { pre[i] = run; if (!F::is_zero(in[i].z)) run = F::mul(run, in[i].z); }
    E inv = F::inv(run);
    for (size_t i = n; i-- > 0;) {
        if (F::is_zero(in[i].z)) { out[i].x = F::zero(); out[i].y = F::zero(); continue; }
        const E zi = F::mul(inv, pre[i]);
        inv = F::mul(inv, in[i].z);
        const E zi2 = F::sqr(zi);
        out[i].x = F::mul(in[i].x, zi2);
        out[i].y = F::mul(in[i].y, F::mul(zi2, zi));
    }
}
}  // namespace

extern "C" int zk_pk_from_bellman_json(const char *path, zk_pk **out) try {
    if (!path || !out) return jfail(ZK_ERR_ARG, "null argument");
    std::string js;
    {
        struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } fc{fopen(path, "rb")};
        FILE *f = fc.f;
        if (!f) return jfail(ZK_ERR_IO, std::string("cannot open ") + path);
        char buf[1 << 16]; size_t k;
        while ((k = fread(buf, 1, sizeof(buf), f)) > 0) js.append(buf, k);
    }
    std::vector<JacG1> A, B1, Cq, Hq, one;
    std::vector<JacG2> B2, two;
    if (!read_g1s(js, "A", A)) return jfail(ZK_ERR_FORMAT, "bellman pk JSON: bad or missing \"A\"");
    if (!read_g1s(js, "B1", B1)) return jfail(ZK_ERR_FORMAT, "bellman pk JSON: bad or missing \"B1\"");
    if (!read_g2s(js, "B2", B2)) return jfail(ZK_ERR_FORMAT, "bellman pk JSON: bad or missing \"B2\"");
    if (!read_g1s(js, "C", Cq)) return jfail(ZK_ERR_FORMAT, "bellman pk JSON: bad or missing \"C\"");
    if (!read_g1s(js, "hExps", Hq)) return jfail(ZK_ERR_FORMAT, "bellman pk JSON: bad or missing \"hExps\"");
    if (B1.size() != B2.size()) return jfail(ZK_ERR_FORMAT, "bellman pk JSON: B1 and B2 differ in length");
    if (B1.size() > A.size()) return jfail(ZK_ERR_FORMAT, "bellman pk JSON: B1 longer than A");
    std::vector<JacG1> singles1(3);
    std::vector<JacG2> singles2(2);
    const char *k1[3] = {"vk_alfa_1", "vk_beta_1", "vk_delta_1"}, *k2[2] = {"vk_beta_2", "vk_delta_2"};
    for (int i = 0; i < 3; i++) { one.clear(); if (!read_g1s(js, k1[i], one) || one.size() != 1) return jfail(ZK_ERR_FORMAT, std::string("bellman pk JSON: bad or missing \"") + k1[i] + "\""); singles1[i] = one[0]; }
    for (int i = 0; i < 2; i++) { two.clear(); if (!read_g2s(js, k2[i], two) || two.size() != 1) return jfail(ZK_ERR_FORMAT, std::string("bellman pk JSON: bad or missing \"") + k2[i] + "\""); singles2[i] = two[0]; }
    js.clear(); js.shrink_to_fit();

    // nozk conversion: drop A's zeros; B keeps G2 where the G1 half is non-zero; L = C[2..]
    std::vector<uint32_t> a_idx, b_idx;
    std::vector<JacG1> a_j, l_j;
    std::vector<JacG2> b_j;
    for (size_t i = 0; i < A.size(); i++) if (!Fq::is_zero(A[i].z)) { a_idx.push_back((uint32_t)i); a_j.push_back(A[i]); }
    for (size_t i = 0; i < B1.size(); i++) if (!Fq::is_zero(B1[i].z)) { b_idx.push_back((uint32_t)i); b_j.push_back(B2[i]); }
    for (size_t i = 2; i < Cq.size(); i++) l_j.push_back(Cq[i]);
    std::vector<G1::Affine> a_val, L, H, s1;
    std::vector<G2::Affine> b_val, s2;
    normalise<Fq, fe>(a_j, a_val); normalise<Fq, fe>(l_j, L); normalise<Fq, fe>(Hq, H); normalise<Fq, fe>(singles1, s1);
    normalise<Fq2, fe2>(b_j, b_val); normalise<Fq2, fe2>(singles2, s2);
    return zk_pk_from_parts((const uint64_t *)&s1[0], (const uint64_t *)&s1[1], (const uint64_t *)&s2[0], (const uint64_t *)&s1[2], (const uint64_t *)&s2[1],
                            (uint32_t)A.size(), (uint32_t)a_idx.size(), a_idx.data(), (const uint64_t *)a_val.data(),
                            (uint32_t)A.size(), (uint32_t)b_idx.size(), b_idx.data(), (const uint64_t *)b_val.data(),
                            (uint32_t)H.size(), (const uint64_t *)H.data(), (uint32_t)L.size(), (const uint64_t *)L.data(), out);
} ZK_GUARD

// pk_bellman2ethsnarks (src/export.cpp:267-328): JSON in, nozk `.raw` out
extern "C" int zk_pk_bellman2ethsnarks(const char *bellman_pk_json, const char *pk_raw) try {
    zk_pk *pk = nullptr;
    int rc = zk_pk_from_bellman_json(bellman_pk_json, &pk);
    if (rc != ZK_OK) return rc;
    rc = zk_pk_save_raw(pk, pk_raw, ZK_CODEC_ALT_BN128);
    zk_pk_free(pk);
    return rc;
} ZK_GUARD
