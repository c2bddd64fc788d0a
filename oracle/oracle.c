/*
 * oracle/oracle.c -- TEST INFRASTRUCTURE: CPU oracle for the Groth16 proving path.  See oracle.h
 * for what it restates (file:line into /root/reference) and how its parity is pinned.
 * Build: make -C oracle   (gcc -O3 -march=native -fopenmp -shared)
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"
#include "field.h"

const fe_t fr_ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
const fe_t fr_R2  = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
const fe_t fq_ONE = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};
const fe_t fq_R2  = {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};
static const fq2_t fq2_ONE = {{{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}}, {{0, 0, 0, 0}}};

/* ---- G1 over Fq */
#define FE fe_t
#define FN(op) fq_##op
#define PN(name) g1_##name
#define JAC g1j_t
#define AFF g1a_t
#include "curve_tmpl.h"
#undef FE
#undef FN
#undef PN
#undef JAC
#undef AFF
/* ---- G2 over Fq2 */
#define FE fq2_t
#define FN(op) fq2_##op
#define PN(name) g2_##name
#define JAC g2j_t
#define AFF g2a_t
#include "curve_tmpl.h"
#undef FE
#undef FN
#undef PN
#undef JAC
#undef AFF

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* =========================================================== bulk field helpers */
void orc_fr_to_mont(uint64_t *o, const uint64_t *in, size_t n) { for (size_t i = 0; i < n; i++) fr_to_mont((fe_t *)(o + 4 * i), (const fe_t *)(in + 4 * i)); }
void orc_fr_from_mont(uint64_t *o, const uint64_t *in, size_t n) { for (size_t i = 0; i < n; i++) fr_from_mont((fe_t *)(o + 4 * i), (const fe_t *)(in + 4 * i)); }
void orc_fq_to_mont(uint64_t *o, const uint64_t *in, size_t n) { for (size_t i = 0; i < n; i++) fq_to_mont((fe_t *)(o + 4 * i), (const fe_t *)(in + 4 * i)); }
void orc_fq_from_mont(uint64_t *o, const uint64_t *in, size_t n) { for (size_t i = 0; i < n; i++) fq_from_mont((fe_t *)(o + 4 * i), (const fe_t *)(in + 4 * i)); }
void orc_fr_mul(uint64_t *o, const uint64_t *a, const uint64_t *b, size_t n) { for (size_t i = 0; i < n; i++) fr_mul((fe_t *)(o + 4 * i), (const fe_t *)(a + 4 * i), (const fe_t *)(b + 4 * i)); }
void orc_fq_mul(uint64_t *o, const uint64_t *a, const uint64_t *b, size_t n) { for (size_t i = 0; i < n; i++) fq_mul((fe_t *)(o + 4 * i), (const fe_t *)(a + 4 * i), (const fe_t *)(b + 4 * i)); }

/* =========================================================== evaluation domain (Appendix A.2) */
/* omega_{2^28} = 5^((r-1)/2^28), canonical; libff alt_bn128 Fr::root_of_unity [ABSENT], verified in SURVEY A.1 */
static const fe_t ROOT_2_28_CANON = {{0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL}};

static void fr_pow_u64(fe_t *r, const fe_t *a, uint64_t e) { uint64_t ee[4] = {e, 0, 0, 0}; fr_pow(r, a, ee); }

static void domain_omega(fe_t *w, uint32_t logm) {
    fe_t t; fr_to_mont(&t, &ROOT_2_28_CANON);
    for (uint32_t i = logm; i < 28; i++) fr_sqr(&t, &t);
    *w = t;
}

static void bitrev_permute(fe_t *a, uint32_t logm) {
    size_t m = (size_t)1 << logm;
    for (size_t i = 0; i < m; i++) {
        size_t j = 0;
        for (uint32_t b = 0; b < logm; b++) j |= ((i >> b) & 1) << (logm - 1 - b);
        if (i < j) { fe_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

/* multiply a[i] by s * g^i */
static void scale_geometric(fe_t *a, size_t m, const fe_t *g, const fe_t *s) {
    const size_t CH = 4096;
    #pragma omp parallel for schedule(static)
    for (size_t c0 = 0; c0 < m; c0 += CH) {
        fe_t t; fr_pow_u64(&t, g, c0); fr_mul(&t, &t, s);
        size_t end = c0 + CH < m ? c0 + CH : m;
        for (size_t i = c0; i < end; i++) { fr_mul(&a[i], &a[i], &t); fr_mul(&t, &t, g); }
    }
}

static void ntt_core(fe_t *a, uint32_t logm, const fe_t *w) {
    size_t m = (size_t)1 << logm;
    if (m == 1) return;
    fe_t *tw = (fe_t *)malloc(sizeof(fe_t) * (m / 2));
    tw[0] = fr_ONE;
    for (size_t i = 1; i < m / 2; i++) fr_mul(&tw[i], &tw[i - 1], w);
    bitrev_permute(a, logm);
    for (uint32_t s = 1; s <= logm; s++) {
        size_t half = (size_t)1 << (s - 1), stride = m >> s;
        #pragma omp parallel for schedule(static)
        for (size_t k = 0; k < m / 2; k++) {
            size_t j = k & (half - 1), base = (k >> (s - 1)) << s;
            fe_t *u = &a[base + j], *v = &a[base + j + half], t;
            fr_mul(&t, v, &tw[j * stride]);
            fr_sub(v, u, &t); fr_add(u, u, &t);
        }
    }
    free(tw);
}

void orc_ntt(uint64_t *data, uint32_t logm, int inverse, int coset) {
    fe_t *a = (fe_t *)data;
    size_t m = (size_t)1 << logm;
    fe_t w, g, gi, mi, t;
    domain_omega(&w, logm);
    fr_set_u64(&g, 5);
    if (!inverse) {
        if (coset) scale_geometric(a, m, &g, &fr_ONE);
        ntt_core(a, logm, &w);
    } else {
        fr_inv(&w, &w);
        ntt_core(a, logm, &w);
        fr_set_u64(&t, m); fr_inv(&mi, &t);
        if (coset) { fr_inv(&gi, &g); scale_geometric(a, m, &gi, &mi); }
        else scale_geometric(a, m, &fr_ONE, &mi);
    }
}

uint32_t orc_domain_size(uint32_t nC, uint32_t nIn) {
    uint32_t v = nC + nIn + 1;        /* src/stubs.cpp:65 */
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;   /* :49-59 */
    return v;
}
static uint32_t ilog2(uint32_t m) { uint32_t k = 0; while ((1u << k) < m) k++; return k; }

static void csr_dot(fe_t *out, const orc_csr *M, const fe_t *w) {
    #pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t j = 0; j < M->n_rows; j++) {
        fe_t acc = {{0, 0, 0, 0}}, t;
        for (uint32_t k = M->row_ptr[j]; k < M->row_ptr[j + 1]; k++) {
            fr_mul(&t, (const fe_t *)(M->coeff + 4 * (size_t)k), &w[M->col[k]]);
            fr_add(&acc, &acc, &t);
        }
        out[j] = acc;
    }
}

/* r1cs_to_qap_witness_map, d1=d2=d3=0 (call site tcc:461-468) */
int orc_witness_map(const orc_r1cs *cs, const uint64_t *witness, uint64_t *h_out) {
    const fe_t *w = (const fe_t *)witness;
    uint32_t m = orc_domain_size(cs->nC, cs->nIn), logm = ilog2(m);
    fe_t *aA = (fe_t *)calloc(m, sizeof(fe_t)), *aB = (fe_t *)calloc(m, sizeof(fe_t)), *aC = (fe_t *)calloc(m, sizeof(fe_t));
    if (!aA || !aB || !aC) return -1;
    csr_dot(aA, &cs->A, w); csr_dot(aB, &cs->B, w); csr_dot(aC, &cs->C, w);
    for (uint32_t i = 0; i <= cs->nIn; i++) aA[cs->nC + i] = w[i];
    orc_ntt((uint64_t *)aA, logm, 1, 0); orc_ntt((uint64_t *)aA, logm, 0, 1);
    orc_ntt((uint64_t *)aB, logm, 1, 0); orc_ntt((uint64_t *)aB, logm, 0, 1);
    orc_ntt((uint64_t *)aC, logm, 1, 0); orc_ntt((uint64_t *)aC, logm, 0, 1);
    fe_t g, zinv;
    fr_set_u64(&g, 5); fr_pow_u64(&zinv, &g, m); fr_sub(&zinv, &zinv, &fr_ONE); fr_inv(&zinv, &zinv);
    #pragma omp parallel for schedule(static)
    for (uint32_t j = 0; j < m; j++) {
        fe_t t; fr_mul(&t, &aA[j], &aB[j]); fr_sub(&t, &t, &aC[j]); fr_mul(&aA[j], &t, &zinv);
    }
    orc_ntt((uint64_t *)aA, logm, 1, 1);
    memcpy(h_out, aA, sizeof(fe_t) * m);
    memset(h_out + 4 * (size_t)m, 0, sizeof(fe_t));
    free(aA); free(aB); free(aC);
    return 0;
}

/* =========================================================== MSM entry points */
static fe_t *scalars_canon(const uint64_t *scalars, size_t n) {
    fe_t *s = (fe_t *)malloc(sizeof(fe_t) * (n ? n : 1));
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) fr_from_mont(&s[i], (const fe_t *)(scalars + 4 * i));
    return s;
}
void orc_msm_g1(const uint64_t *bases, const uint64_t *scalars, size_t n, unsigned c, uint64_t out[8]) {
    fe_t *s = scalars_canon(scalars, n); g1j_t r;
    g1_msm(&r, (const g1a_t *)bases, s, n, &fq_ONE, c); g1_to_aff((g1a_t *)out, &r); free(s);
}
void orc_msm_g2(const uint64_t *bases, const uint64_t *scalars, size_t n, unsigned c, uint64_t out[16]) {
    fe_t *s = scalars_canon(scalars, n); g2j_t r;
    g2_msm(&r, (const g2a_t *)bases, s, n, &fq2_ONE, c); g2_to_aff((g2a_t *)out, &r); free(s);
}
void orc_msm_g1_naive(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out[8]) {
    fe_t *s = scalars_canon(scalars, n); g1j_t r;
    g1_msm_naive(&r, (const g1a_t *)bases, s, n, &fq_ONE); g1_to_aff((g1a_t *)out, &r); free(s);
}
void orc_msm_g2_naive(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out[16]) {
    fe_t *s = scalars_canon(scalars, n); g2j_t r;
    g2_msm_naive(&r, (const g2a_t *)bases, s, n, &fq2_ONE); g2_to_aff((g2a_t *)out, &r); free(s);
}

/* generators: G1 (1,2); G2 standard alt_bn128 generator, canonical */
static const uint64_t G2_GEN_CANON[16] = {
    0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL,   /* x.c0 */
    0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL,   /* x.c1 */
    0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL,   /* y.c0 */
    0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};  /* y.c1 */

static void g1_generator(g1j_t *g) { fq_set_u64(&g->X, 1); fq_set_u64(&g->Y, 2); g->Z = fq_ONE; }
static void g2_generator(g2j_t *g) {
    fe_t t[4]; orc_fq_to_mont((uint64_t *)t, G2_GEN_CANON, 4);
    g->X.c0 = t[0]; g->X.c1 = t[1]; g->Y.c0 = t[2]; g->Y.c1 = t[3]; g->Z = fq2_ONE;
}
void orc_batch_mul_g1(const uint64_t *scalars, size_t n, uint64_t *out) {
    fe_t *s = scalars_canon(scalars, n); g1j_t g; g1_generator(&g);
    g1_batch_mul((g1a_t *)out, &g, s, n, &fq_ONE); free(s);
}
void orc_batch_mul_g2(const uint64_t *scalars, size_t n, uint64_t *out) {
    fe_t *s = scalars_canon(scalars, n); g2j_t g; g2_generator(&g);
    g2_batch_mul((g2a_t *)out, &g, s, n, &fq2_ONE); free(s);
}

/* =========================================================== keys */
struct orc_pk {
    g1a_t alpha_g1, beta_g1, delta_g1; g2a_t beta_g2, delta_g2;
    uint32_t a_domain, nA; uint32_t *a_idx; g1a_t *a_val;
    uint32_t b_domain, nB; uint32_t *b_idx; g2a_t *b_val;
    uint32_t nH; g1a_t *H; uint32_t nL; g1a_t *L;
};
struct orc_vk { g1a_t alpha_g1; g2a_t beta_g2, gamma_g2, delta_g2; uint32_t n_abc; g1a_t *gamma_abc; };

void orc_pk_free(orc_pk *pk) { if (!pk) return; free(pk->a_idx); free(pk->a_val); free(pk->b_idx); free(pk->b_val); free(pk->H); free(pk->L); free(pk); }
void orc_vk_free(orc_vk *vk) { if (!vk) return; free(vk->gamma_abc); free(vk); }
void orc_pk_sizes(const orc_pk *pk, uint32_t s[6]) { s[0] = pk->a_domain; s[1] = pk->nA; s[2] = pk->b_domain; s[3] = pk->nB; s[4] = pk->nH; s[5] = pk->nL; }
const void *orc_pk_ptr(const orc_pk *pk, int which) {
    switch (which) {
    case 0: return &pk->alpha_g1; case 1: return &pk->beta_g1; case 2: return &pk->beta_g2;
    case 3: return &pk->delta_g1; case 4: return &pk->delta_g2;
    case 5: return pk->a_idx; case 6: return pk->a_val; case 7: return pk->b_idx; case 8: return pk->b_val;
    case 9: return pk->H; case 10: return pk->L; default: return NULL; }
}
static void *dup_mem(const void *p, size_t n) { void *r = malloc(n ? n : 1); if (n) memcpy(r, p, n); return r; }
int orc_pk_from_parts(const uint64_t *alpha_g1, const uint64_t *beta_g1, const uint64_t *beta_g2,
                      const uint64_t *delta_g1, const uint64_t *delta_g2,
                      uint32_t a_domain, uint32_t nA, const uint32_t *a_idx, const uint64_t *a_val,
                      uint32_t b_domain, uint32_t nB, const uint32_t *b_idx, const uint64_t *b_val,
                      uint32_t nH, const uint64_t *H, uint32_t nL, const uint64_t *L, orc_pk **out) {
    orc_pk *pk = (orc_pk *)calloc(1, sizeof(orc_pk));
    memcpy(&pk->alpha_g1, alpha_g1, 64); memcpy(&pk->beta_g1, beta_g1, 64); memcpy(&pk->beta_g2, beta_g2, 128);
    memcpy(&pk->delta_g1, delta_g1, 64); memcpy(&pk->delta_g2, delta_g2, 128);
    pk->a_domain = a_domain; pk->nA = nA; pk->a_idx = dup_mem(a_idx, 4 * (size_t)nA); pk->a_val = dup_mem(a_val, 64 * (size_t)nA);
    pk->b_domain = b_domain; pk->nB = nB; pk->b_idx = dup_mem(b_idx, 4 * (size_t)nB); pk->b_val = dup_mem(b_val, 128 * (size_t)nB);
    pk->nH = nH; pk->H = dup_mem(H, 64 * (size_t)nH); pk->nL = nL; pk->L = dup_mem(L, 64 * (size_t)nL);
    *out = pk; return 0;
}

static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static void draw_fr_canon(uint64_t out[4], uint64_t *state) {
    for (int i = 0; i < 4; i++) out[i] = splitmix64(state);
    while (limbs_geq(out, FR_P)) limbs_sub(out, out, FR_P);
}

/* batch inversion (Montgomery's trick); zeros stay zero */
static void fr_batch_inv(fe_t *a, size_t n) {
    fe_t *pre = (fe_t *)malloc(sizeof(fe_t) * (n ? n : 1)), acc = fr_ONE;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!fr_is_zero(&a[i])) fr_mul(&acc, &acc, &a[i]); }
    fr_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fr_is_zero(&a[i])) continue;
        fe_t t; fr_mul(&t, &acc, &pre[i]); fr_mul(&acc, &acc, &a[i]); a[i] = t;
    }
    free(pre);
}

/* r1cs_gg_ppzksnark_zok_generator tcc:277-449 with r1cs_to_qap_instance_map_with_evaluation
 * (Appendix A.4) and the zk -> nozk key conversion hpp:209-233 */
int orc_keygen_explicit(const orc_r1cs *cs, const uint64_t toxic[20], orc_pk **pk_out, orc_vk **vk_out) {
    uint32_t nC = cs->nC, nIn = cs->nIn, V = cs->V;
    uint32_t m = orc_domain_size(nC, nIn), logm = ilog2(m);
    fe_t t, alpha, beta, gamma, delta;
    fr_to_mont(&t, (const fe_t *)(toxic + 0)); fr_to_mont(&alpha, (const fe_t *)(toxic + 4)); fr_to_mont(&beta, (const fe_t *)(toxic + 8));
    fr_to_mont(&gamma, (const fe_t *)(toxic + 12)); fr_to_mont(&delta, (const fe_t *)(toxic + 16));
    /* Lagrange basis at t: u_i = omega^i (t^m - 1) / (m (t - omega^i)) */
    fe_t w, Zt, mfe, wi = fr_ONE;
    domain_omega(&w, logm);
    fr_pow_u64(&Zt, &t, m); fr_sub(&Zt, &Zt, &fr_ONE);
    fr_set_u64(&mfe, m);
    fe_t *u = (fe_t *)malloc(sizeof(fe_t) * m), *den = (fe_t *)malloc(sizeof(fe_t) * m);
    for (uint32_t i = 0; i < m; i++) {
        fe_t d; fr_sub(&d, &t, &wi); fr_mul(&den[i], &d, &mfe);
        fr_mul(&u[i], &wi, &Zt); fr_mul(&wi, &wi, &w);
    }
    fr_batch_inv(den, m);
    for (uint32_t i = 0; i < m; i++) fr_mul(&u[i], &u[i], &den[i]);
    free(den);
    fe_t *At = (fe_t *)calloc(V + 1, sizeof(fe_t)), *Bt = (fe_t *)calloc(V + 1, sizeof(fe_t)), *Ct = (fe_t *)calloc(V + 1, sizeof(fe_t));
    for (uint32_t i = 0; i <= nIn; i++) At[i] = u[nC + i];
    const orc_csr *Ms[3] = {&cs->A, &cs->B, &cs->C}; fe_t *Ts[3] = {At, Bt, Ct};
    for (int q = 0; q < 3; q++)
        for (uint32_t j = 0; j < nC; j++)
            for (uint32_t k = Ms[q]->row_ptr[j]; k < Ms[q]->row_ptr[j + 1]; k++) {
                fe_t p; fr_mul(&p, &u[j], (const fe_t *)(Ms[q]->coeff + 4 * (size_t)k));
                fr_add(&Ts[q][Ms[q]->col[k]], &Ts[q][Ms[q]->col[k]], &p);
            }
    free(u);
    fe_t gi, di;
    fr_inv(&gi, &gamma); fr_inv(&di, &delta);
    fe_t *abc = (fe_t *)malloc(sizeof(fe_t) * (V + 1));          /* (beta At + alpha Bt + Ct) / {gamma|delta}, tcc:326-342 */
    for (uint32_t i = 0; i <= V; i++) {
        fe_t a, b; fr_mul(&a, &beta, &At[i]); fr_mul(&b, &alpha, &Bt[i]); fr_add(&a, &a, &b); fr_add(&a, &a, &Ct[i]);
        fr_mul(&abc[i], &a, i <= nIn ? &gi : &di);
    }
    fe_t *Hs = (fe_t *)malloc(sizeof(fe_t) * (m - 1));           /* t^j Zt / delta, j < m-1 (tcc:350,400) */
    fe_t zd, tj = fr_ONE; fr_mul(&zd, &Zt, &di);
    for (uint32_t j = 0; j + 1 < m; j++) { fr_mul(&Hs[j], &tj, &zd); fr_mul(&tj, &tj, &t); }

    orc_pk *pk = (orc_pk *)calloc(1, sizeof(orc_pk));
    orc_vk *vk = (orc_vk *)calloc(1, sizeof(orc_vk));
    fe_t single[3] = {alpha, beta, delta}; g1a_t s1[3];
    orc_batch_mul_g1((const uint64_t *)single, 3, (uint64_t *)s1);
    pk->alpha_g1 = s1[0]; pk->beta_g1 = s1[1]; pk->delta_g1 = s1[2];
    fe_t single2[3] = {beta, gamma, delta}; g2a_t s2[3];
    orc_batch_mul_g2((const uint64_t *)single2, 3, (uint64_t *)s2);
    pk->beta_g2 = s2[0]; pk->delta_g2 = s2[2];
    vk->alpha_g1 = s1[0]; vk->beta_g2 = s2[0]; vk->gamma_g2 = s2[1]; vk->delta_g2 = s2[2];
    vk->n_abc = nIn + 1; vk->gamma_abc = (g1a_t *)malloc(sizeof(g1a_t) * (nIn + 1));
    orc_batch_mul_g1((const uint64_t *)abc, nIn + 1, (uint64_t *)vk->gamma_abc);
    /* A query: zero entries dropped (hpp:216-224) */
    pk->a_domain = V + 1; pk->b_domain = V + 1;
    pk->a_idx = (uint32_t *)malloc(4 * (size_t)(V + 1)); pk->b_idx = (uint32_t *)malloc(4 * (size_t)(V + 1));
    fe_t *as = (fe_t *)malloc(sizeof(fe_t) * (V + 1)), *bs = (fe_t *)malloc(sizeof(fe_t) * (V + 1));
    for (uint32_t i = 0; i <= V; i++) {
        if (!fr_is_zero(&At[i])) { pk->a_idx[pk->nA] = i; as[pk->nA++] = At[i]; }
        if (!fr_is_zero(&Bt[i])) { pk->b_idx[pk->nB] = i; bs[pk->nB++] = Bt[i]; }
    }
    pk->a_val = (g1a_t *)malloc(sizeof(g1a_t) * (pk->nA ? pk->nA : 1));
    pk->b_val = (g2a_t *)malloc(sizeof(g2a_t) * (pk->nB ? pk->nB : 1));
    orc_batch_mul_g1((const uint64_t *)as, pk->nA, (uint64_t *)pk->a_val);
    orc_batch_mul_g2((const uint64_t *)bs, pk->nB, (uint64_t *)pk->b_val);
    pk->nH = m - 1; pk->H = (g1a_t *)malloc(sizeof(g1a_t) * (m - 1));
    orc_batch_mul_g1((const uint64_t *)Hs, m - 1, (uint64_t *)pk->H);
    pk->nL = V - nIn; pk->L = (g1a_t *)malloc(sizeof(g1a_t) * (pk->nL ? pk->nL : 1));
    orc_batch_mul_g1((const uint64_t *)(abc + nIn + 1), pk->nL, (uint64_t *)pk->L);
    free(At); free(Bt); free(Ct); free(abc); free(Hs); free(as); free(bs);
    *pk_out = pk; *vk_out = vk;
    return 0;
}

int orc_keygen(const orc_r1cs *cs, uint64_t seed, orc_pk **pk_out, orc_vk **vk_out) {
    uint64_t toxic[20], st = seed;
    for (int i = 0; i < 5; i++) draw_fr_canon(toxic + 4 * i, &st);
    return orc_keygen_explicit(cs, toxic, pk_out, vk_out);
}

static void g1_canon(const g1a_t *p, uint64_t x[4], uint64_t y[4], uint32_t *inf);
static void g2_canon(const g2a_t *p, uint64_t xc0[4], uint64_t xc1[4], uint64_t yc0[4], uint64_t yc1[4], uint32_t *inf);
/* =========================================================== closed form from the toxic waste
 * The no-ZK proof straight from the definitions (the reference's comments at tcc:533-540 with the key elements of
 * tcc:326-350 and the QAP of SURVEY Appendix A.3 / A.4):
 *   A = (alpha + sum_i w_i A_i(t)) G1,  B = (beta + sum_i w_i B_i(t)) G2,
 *   C = ((sum_{i>nIn} w_i (beta A_i + alpha B_i + C_i)(t) + A(t) B(t) - C(t)) / delta) G1     (H(t) Z(t) = A(t) B(t) - C(t)).
 * O(nnz) field operations and THREE plain double-and-add scalar multiplications: no multi-exponentiation, no transform,
 * no proving key, no stream codec is touched -- an independent pin of orc_prove and of the HIP path. */
int orc_proof_from_trapdoor(const orc_r1cs *cs, const uint64_t *witness, const uint64_t toxic[20], orc_proof *out) {
    const fe_t *w = (const fe_t *)witness;
    uint32_t nC = cs->nC, nIn = cs->nIn;
    uint32_t m = orc_domain_size(nC, nIn), logm = ilog2(m);
    fe_t t, alpha, beta, delta;
    fr_to_mont(&t, (const fe_t *)(toxic + 0)); fr_to_mont(&alpha, (const fe_t *)(toxic + 4)); fr_to_mont(&beta, (const fe_t *)(toxic + 8));
    fr_to_mont(&delta, (const fe_t *)(toxic + 16));
    fe_t om, Zt, mfe, wi = fr_ONE;
    domain_omega(&om, logm);
    fr_pow_u64(&Zt, &t, m); fr_sub(&Zt, &Zt, &fr_ONE);
    fr_set_u64(&mfe, m);
    fe_t *u = (fe_t *)malloc(sizeof(fe_t) * m), *den = (fe_t *)malloc(sizeof(fe_t) * m);
    if (!u || !den) { free(u); free(den); return -1; }
    for (uint32_t j = 0; j < m; j++) {                      /* u_j = omega^j Z(t) / (m (t - omega^j)) */
        fe_t d; fr_sub(&d, &t, &wi); fr_mul(&den[j], &d, &mfe);
        fr_mul(&u[j], &wi, &Zt); fr_mul(&wi, &wi, &om);
    }
    fr_batch_inv(den, m);
    for (uint32_t j = 0; j < m; j++) fr_mul(&u[j], &u[j], &den[j]);
    free(den);
    const orc_csr *Ms[3] = {&cs->A, &cs->B, &cs->C};
    fe_t S[3], P[3];                                        /* sums over all variables / over the public ones (i <= nIn) */
    for (int q = 0; q < 3; q++) {
        fe_t tot = {{0, 0, 0, 0}}, pub = {{0, 0, 0, 0}};
        for (uint32_t j = 0; j < nC; j++) {
            fe_t row = {{0, 0, 0, 0}}, prow = {{0, 0, 0, 0}}, v;
            for (uint32_t k = Ms[q]->row_ptr[j]; k < Ms[q]->row_ptr[j + 1]; k++) {
                fr_mul(&v, (const fe_t *)(Ms[q]->coeff + 4 * (size_t)k), &w[Ms[q]->col[k]]);
                fr_add(&row, &row, &v);
                if (Ms[q]->col[k] <= nIn) fr_add(&prow, &prow, &v);
            }
            fr_mul(&v, &row, &u[j]); fr_add(&tot, &tot, &v);
            fr_mul(&v, &prow, &u[j]); fr_add(&pub, &pub, &v);
        }
        S[q] = tot; P[q] = pub;
    }
    for (uint32_t i = 0; i <= nIn; i++) {                   /* input-consistency rows of A */
        fe_t v; fr_mul(&v, &u[nC + i], &w[i]);
        fr_add(&S[0], &S[0], &v); fr_add(&P[0], &P[0], &v);
    }
    free(u);
    fe_t a, b, c, x, y, di;
    fr_add(&a, &alpha, &S[0]); fr_add(&b, &beta, &S[1]);
    fr_sub(&x, &S[0], &P[0]); fr_mul(&c, &beta, &x);
    fr_sub(&x, &S[1], &P[1]); fr_mul(&y, &alpha, &x); fr_add(&c, &c, &y);
    fr_sub(&x, &S[2], &P[2]); fr_add(&c, &c, &x);
    fr_mul(&y, &S[0], &S[1]); fr_add(&c, &c, &y); fr_sub(&c, &c, &S[2]);
    fr_inv(&di, &delta); fr_mul(&c, &c, &di);
    fe_t ka, kb, kc;
    fr_from_mont(&ka, &a); fr_from_mont(&kb, &b); fr_from_mont(&kc, &c);
    g1j_t g1, ja, jc; g2j_t g2, jb; g1a_t ra, rc; g2a_t rb;
    g1_generator(&g1); g2_generator(&g2);
    g1_mul_scalar(&ja, &g1, ka.l); g2_mul_scalar(&jb, &g2, kb.l); g1_mul_scalar(&jc, &g1, kc.l);
    g1_to_aff(&ra, &ja); g2_to_aff(&rb, &jb); g1_to_aff(&rc, &jc);
    memset(out, 0, sizeof *out);
    g1_canon(&ra, out->a_x, out->a_y, &out->a_inf);
    g2_canon(&rb, out->b_x_c0, out->b_x_c1, out->b_y_c0, out->b_y_c1, &out->b_inf);
    g1_canon(&rc, out->c_x, out->c_y, &out->c_inf);
    return 0;
}

/* =========================================================== .raw stream (tcc:108-143, utils.hpp:166-185) */
/* upstream libff encoding under BINARY_OUTPUT + MONTGOMERY_OUTPUT + NO_PT_COMPRESSION:
 * point = ASCII '0'/'1' infinity flag, then raw Montgomery limbs of X, Y (G2: X.c0 X.c1 Y.c0 Y.c1);
 * infinity is written with affine X = 0, Y = 1; OUTPUT_NEWLINE / OUTPUT_SEPARATOR are empty. */
static void raw_put_g1(FILE *f, const g1a_t *p) {
    if (g1_aff_is_inf(p)) { fe_t z = {{0, 0, 0, 0}}; fputc('1', f); fwrite(&z, 32, 1, f); fwrite(&fq_ONE, 32, 1, f); }
    else { fputc('0', f); fwrite(p, 64, 1, f); }
}
static void raw_put_g2(FILE *f, const g2a_t *p) {
    if (g2_aff_is_inf(p)) { fq2_t z; memset(&z, 0, sizeof z); fputc('1', f); fwrite(&z, 64, 1, f); fwrite(&fq2_ONE, 64, 1, f); }
    else { fputc('0', f); fwrite(p, 128, 1, f); }
}
static int raw_get_g1(FILE *f, g1a_t *p) {
    int c = fgetc(f); if (c != '0' && c != '1') return -1;
    if (fread(p, 64, 1, f) != 1) return -1;
    if (c == '1') memset(p, 0, sizeof *p);
    return 0;
}
static int raw_get_g2(FILE *f, g2a_t *p) {
    int c = fgetc(f); if (c != '0' && c != '1') return -1;
    if (fread(p, 128, 1, f) != 1) return -1;
    if (c == '1') memset(p, 0, sizeof *p);
    return 0;
}
static int raw_get_size(FILE *f, size_t *v) { unsigned long long x; if (fscanf(f, "%llu", &x) != 1) return -1; if (fgetc(f) != '\n') return -1; *v = x; return 0; }

int orc_pk_write_raw(const orc_pk *pk, const char *path) {
    FILE *f = fopen(path, "wb"); if (!f) return -1;
    raw_put_g1(f, &pk->alpha_g1); raw_put_g1(f, &pk->beta_g1); raw_put_g2(f, &pk->beta_g2);
    raw_put_g1(f, &pk->delta_g1); raw_put_g2(f, &pk->delta_g2);
    fprintf(f, "%u\n%u\n", pk->a_domain, pk->nA);
    for (uint32_t i = 0; i < pk->nA; i++) fprintf(f, "%u\n", pk->a_idx[i]);
    fprintf(f, "%u\n", pk->nA);
    for (uint32_t i = 0; i < pk->nA; i++) raw_put_g1(f, &pk->a_val[i]);
    fprintf(f, "%u\n%u\n", pk->b_domain, pk->nB);
    for (uint32_t i = 0; i < pk->nB; i++) fprintf(f, "%u\n", pk->b_idx[i]);
    fprintf(f, "%u\n", pk->nB);
    for (uint32_t i = 0; i < pk->nB; i++) raw_put_g2(f, &pk->b_val[i]);
    fprintf(f, "%u\n", pk->nH);
    for (uint32_t i = 0; i < pk->nH; i++) raw_put_g1(f, &pk->H[i]);
    fprintf(f, "%u\n", pk->nL);
    for (uint32_t i = 0; i < pk->nL; i++) raw_put_g1(f, &pk->L[i]);
    int rc = ferror(f) ? -1 : 0;
    fclose(f); return rc;
}

int orc_pk_read_raw(const char *path, orc_pk **out) {
    FILE *f = fopen(path, "rb"); if (!f) return -1;
    orc_pk *pk = (orc_pk *)calloc(1, sizeof(orc_pk)); size_t n, nv;
    int bad = 0;
    bad |= raw_get_g1(f, &pk->alpha_g1) | raw_get_g1(f, &pk->beta_g1) | raw_get_g2(f, &pk->beta_g2);
    bad |= raw_get_g1(f, &pk->delta_g1) | raw_get_g2(f, &pk->delta_g2);
    if (!bad && !(bad = raw_get_size(f, &n))) pk->a_domain = (uint32_t)n;
    if (!bad && !(bad = raw_get_size(f, &n))) { pk->nA = (uint32_t)n; pk->a_idx = (uint32_t *)malloc(4 * n + 4);
        for (size_t i = 0; i < n && !bad; i++) { size_t v = 0; bad = raw_get_size(f, &v); pk->a_idx[i] = (uint32_t)v; } }
    if (!bad && !(bad = raw_get_size(f, &nv))) { bad = nv != pk->nA; pk->a_val = (g1a_t *)malloc(64 * nv + 64);
        for (size_t i = 0; i < nv && !bad; i++) bad = raw_get_g1(f, &pk->a_val[i]); }
    if (!bad && !(bad = raw_get_size(f, &n))) pk->b_domain = (uint32_t)n;
    if (!bad && !(bad = raw_get_size(f, &n))) { pk->nB = (uint32_t)n; pk->b_idx = (uint32_t *)malloc(4 * n + 4);
        for (size_t i = 0; i < n && !bad; i++) { size_t v = 0; bad = raw_get_size(f, &v); pk->b_idx[i] = (uint32_t)v; } }
    if (!bad && !(bad = raw_get_size(f, &nv))) { bad = nv != pk->nB; pk->b_val = (g2a_t *)malloc(128 * nv + 128);
        for (size_t i = 0; i < nv && !bad; i++) bad = raw_get_g2(f, &pk->b_val[i]); }
    if (!bad && !(bad = raw_get_size(f, &n))) { pk->nH = (uint32_t)n; pk->H = (g1a_t *)malloc(64 * n + 64);
        for (size_t i = 0; i < n && !bad; i++) bad = raw_get_g1(f, &pk->H[i]); }
    if (!bad && !(bad = raw_get_size(f, &n))) { pk->nL = (uint32_t)n; pk->L = (g1a_t *)malloc(64 * n + 64);
        for (size_t i = 0; i < n && !bad; i++) bad = raw_get_g1(f, &pk->L[i]); }
    fclose(f);
    if (bad) { orc_pk_free(pk); return -2; }
    *out = pk; return 0;
}

/* =========================================================== JSON (src/export.cpp:20-145) */
/* HexStringFromBigint: mpz_get_str(., 16, .) -- lowercase, no leading zeros, zero -> "0" */
static size_t hex_canon(char *dst, const uint64_t v[4]) {
    char tmp[65]; int n = 0, started = 0;
    for (int i = 3; i >= 0; i--) for (int s = 60; s >= 0; s -= 4) {
        unsigned d = (unsigned)(v[i] >> s) & 15;
        if (d || started) { tmp[n++] = "0123456789abcdef"[d]; started = 1; }
    }
    if (!n) tmp[n++] = '0';
    memcpy(dst, tmp, n); return n;
}
typedef struct { char *buf; size_t cap, len; } sbuf;
static void sb_put(sbuf *s, const char *t, size_t n) { if (s->len + n < s->cap) memcpy(s->buf + s->len, t, n); s->len += n; }
static void sb_str(sbuf *s, const char *t) { sb_put(s, t, strlen(t)); }
static void sb_hex(sbuf *s, const uint64_t v[4]) { char t[64]; size_t n = hex_canon(t, v); sb_str(s, "\"0x"); sb_put(s, t, n); sb_str(s, "\""); }
static void sb_g1(sbuf *s, const uint64_t x[4], const uint64_t y[4]) { sb_hex(s, x); sb_str(s, ", "); sb_hex(s, y); }
static void sb_g2(sbuf *s, const uint64_t xc0[4], const uint64_t xc1[4], const uint64_t yc0[4], const uint64_t yc1[4]) {
    sb_str(s, "["); sb_hex(s, xc1); sb_str(s, ", "); sb_hex(s, xc0); sb_str(s, "],\n [");
    sb_hex(s, yc1); sb_str(s, ", "); sb_hex(s, yc0); sb_str(s, "]");
}
size_t orc_proof_to_json(const orc_proof *p, const uint64_t *inputs, uint32_t nIn, char *buf, size_t cap) {
    sbuf s = {buf, cap, 0};
    sb_str(&s, "{\n \"A\" :["); sb_g1(&s, p->a_x, p->a_y);
    sb_str(&s, "],\n \"B\"  :["); sb_g2(&s, p->b_x_c0, p->b_x_c1, p->b_y_c0, p->b_y_c1);
    sb_str(&s, "],\n \"C\"  :["); sb_g1(&s, p->c_x, p->c_y);
    sb_str(&s, "],\n \"input\" :[");
    for (uint32_t i = 0; i < nIn; i++) {
        fe_t c; fr_from_mont(&c, (const fe_t *)(inputs + 4 * (size_t)i));
        sb_hex(&s, c.l); if (i + 1 < nIn) sb_str(&s, ", ");
    }
    sb_str(&s, "]\n}");
    if (cap) buf[s.len < cap ? s.len : cap - 1] = 0;
    return s.len;
}
static void g1_canon(const g1a_t *p, uint64_t x[4], uint64_t y[4], uint32_t *inf) {
    if (g1_aff_is_inf(p)) { memset(x, 0, 32); memset(y, 0, 32); y[0] = 1; if (inf) *inf = 1; return; }
    fe_t t; fq_from_mont(&t, &p->x); memcpy(x, t.l, 32); fq_from_mont(&t, &p->y); memcpy(y, t.l, 32); if (inf) *inf = 0;
}
static void g2_canon(const g2a_t *p, uint64_t xc0[4], uint64_t xc1[4], uint64_t yc0[4], uint64_t yc1[4], uint32_t *inf) {
    if (g2_aff_is_inf(p)) { memset(xc0, 0, 32); memset(xc1, 0, 32); memset(yc0, 0, 32); memset(yc1, 0, 32); yc0[0] = 1; if (inf) *inf = 1; return; }
    fe_t t;
    fq_from_mont(&t, &p->x.c0); memcpy(xc0, t.l, 32); fq_from_mont(&t, &p->x.c1); memcpy(xc1, t.l, 32);
    fq_from_mont(&t, &p->y.c0); memcpy(yc0, t.l, 32); fq_from_mont(&t, &p->y.c1); memcpy(yc1, t.l, 32);
    if (inf) *inf = 0;
}
size_t orc_vk_to_json(const orc_vk *vk, char *buf, size_t cap) {
    sbuf s = {buf, cap, 0}; uint64_t a[4], b[4], c[4], d[4];
    sb_str(&s, "{\n \"alpha\" :["); g1_canon(&vk->alpha_g1, a, b, NULL); sb_g1(&s, a, b);
    sb_str(&s, "],\n \"beta\"  :["); g2_canon(&vk->beta_g2, a, b, c, d, NULL); sb_g2(&s, a, b, c, d);
    sb_str(&s, "],\n \"gamma\" :["); g2_canon(&vk->gamma_g2, a, b, c, d, NULL); sb_g2(&s, a, b, c, d);
    sb_str(&s, "],\n \"delta\" :["); g2_canon(&vk->delta_g2, a, b, c, d, NULL); sb_g2(&s, a, b, c, d);
    sb_str(&s, "],\n\"gammaABC\" :[[");
    for (uint32_t i = 0; i < vk->n_abc; i++) {
        if (i) sb_str(&s, ",[");
        g1_canon(&vk->gamma_abc[i], a, b, NULL); sb_g1(&s, a, b); sb_str(&s, "]");
    }
    sb_str(&s, "]}");
    if (cap) buf[s.len < cap ? s.len : cap - 1] = 0;
    return s.len;
}

/* =========================================================== prover (tcc:451-550) */
int orc_prove(const orc_pk *pk, const orc_r1cs *cs, const uint64_t *witness, unsigned msm_c, orc_proof *out, double ph[6]) {
    const fe_t *w = (const fe_t *)witness;
    uint32_t m = orc_domain_size(cs->nC, cs->nIn), V = cs->V, nIn = cs->nIn;
    if (pk->a_domain != V + 1 || pk->b_domain != V + 1 || pk->nH != m - 1 || pk->nL != V - nIn) return -3;   /* tcc:478-482 */
    double t0 = now_s(), t1;
    fe_t *aH = (fe_t *)malloc(sizeof(fe_t) * ((size_t)m + 1));
    if (orc_witness_map(cs, witness, (uint64_t *)aH)) return -1;
    if (!fr_is_zero(&aH[m - 1]) || !fr_is_zero(&aH[m])) { free(aH); return -4; }     /* tcc:472-474 (degree) */
    t1 = now_s(); if (ph) ph[0] = t1 - t0; t0 = t1;
    /* A-query: kc_multi_exp_with_mixed_addition over the sparse index list, tcc:488-495 */
    fe_t *sa = (fe_t *)malloc(sizeof(fe_t) * (pk->nA ? pk->nA : 1));
    for (uint32_t k = 0; k < pk->nA; k++) sa[k] = w[pk->a_idx[k]];
    g1a_t At; orc_msm_g1((const uint64_t *)pk->a_val, (const uint64_t *)sa, pk->nA, msm_c, (uint64_t *)&At);
    free(sa);
    t1 = now_s(); if (ph) ph[1] = t1 - t0; t0 = t1;
    fe_t *sb = (fe_t *)malloc(sizeof(fe_t) * (pk->nB ? pk->nB : 1));
    for (uint32_t k = 0; k < pk->nB; k++) sb[k] = w[pk->b_idx[k]];
    g2a_t Bt; orc_msm_g2((const uint64_t *)pk->b_val, (const uint64_t *)sb, pk->nB, msm_c, (uint64_t *)&Bt);
    free(sb);
    t1 = now_s(); if (ph) ph[2] = t1 - t0; t0 = t1;
    g1a_t Ht; orc_msm_g1((const uint64_t *)pk->H, (const uint64_t *)aH, m - 1, msm_c, (uint64_t *)&Ht);   /* tcc:510-518 */
    free(aH);
    t1 = now_s(); if (ph) ph[3] = t1 - t0; t0 = t1;
    g1a_t Lt; orc_msm_g1((const uint64_t *)pk->L, (const uint64_t *)(w + nIn + 1), pk->nL, msm_c, (uint64_t *)&Lt);  /* tcc:522-530 */
    t1 = now_s(); if (ph) ph[4] = t1 - t0; t0 = t1;
    /* A = alpha + At; B = beta + Bt; C = Ht + Lt  (tcc:533-540) */
    g1j_t ja, jc; g2j_t jb; g1a_t ra, rc; g2a_t rb;
    g1_from_aff(&ja, &pk->alpha_g1, &fq_ONE); g1_madd(&ja, &ja, &At, &fq_ONE); g1_to_aff(&ra, &ja);
    g2_from_aff(&jb, &pk->beta_g2, &fq2_ONE); g2_madd(&jb, &jb, &Bt, &fq2_ONE); g2_to_aff(&rb, &jb);
    g1_from_aff(&jc, &Ht, &fq_ONE); g1_madd(&jc, &jc, &Lt, &fq_ONE); g1_to_aff(&rc, &jc);
    memset(out, 0, sizeof *out);
    g1_canon(&ra, out->a_x, out->a_y, &out->a_inf);
    g2_canon(&rb, out->b_x_c0, out->b_x_c1, out->b_y_c0, out->b_y_c1, &out->b_inf);
    g1_canon(&rc, out->c_x, out->c_y, &out->c_inf);
    t1 = now_s(); if (ph) ph[5] = t1 - t0;
    return 0;
}
