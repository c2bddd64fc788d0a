/*
 * oracle/field.h -- TEST INFRASTRUCTURE (CPU oracle), NOT PRODUCT CODE.
 *
 * Plain-C restatement of the prime-field arithmetic the reference gets from libff
 * (depends/libsnark/depends/libff, ABSENT from /root/reference; CortexFoundation/libsnark@opt).
 * Element = 4 x u64 little-endian limbs in Montgomery form, R = 2^256 -- the in-memory form of
 * libff::Fp_model<4> and the byte layout of the .raw proving key (CMakeLists.txt:115-131:
 * BINARY_OUTPUT + MONTGOMERY_OUTPUT).  Moduli: contracts/Verifier.sol:10,17.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/.
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe_t;

/* r = 21888242871839275222246405745257275088548364400416034343698204186575808495617 */
static const uint64_t FR_P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
#define FR_INV 0xc2e1f593efffffffULL
/* q = 21888242871839275222246405745257275088696311157297823662689037894645226208583 */
static const uint64_t FQ_P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
#define FQ_INV 0x87d20782e4866389ULL

#define ORC_INLINE static inline __attribute__((always_inline))

ORC_INLINE int limbs_geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
ORC_INLINE uint64_t limbs_sub(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
ORC_INLINE uint64_t limbs_add(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a[i] + b[i] + carry;
        r[i] = (uint64_t)s; carry = (uint64_t)(s >> 64);
    }
    return carry;
}

/* Montgomery product a*b*R^-1 mod p, CIOS (Koc et al.), inputs and output fully reduced. */
ORC_INLINE void mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4], const uint64_t p[4], uint64_t inv) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    for (int i = 0; i < 4; i++) {
        u128 c;
        uint64_t bi = b[i];
        c = (u128)a[0] * bi + t0; t0 = (uint64_t)c; c >>= 64;
        c += (u128)a[1] * bi + t1; t1 = (uint64_t)c; c >>= 64;
        c += (u128)a[2] * bi + t2; t2 = (uint64_t)c; c >>= 64;
        c += (u128)a[3] * bi + t3; t3 = (uint64_t)c; c >>= 64;
        c += t4; t4 = (uint64_t)c; uint64_t t5 = (uint64_t)(c >> 64);
        uint64_t m = t0 * inv;
        c = (u128)m * p[0] + t0; c >>= 64;
        c += (u128)m * p[1] + t1; t0 = (uint64_t)c; c >>= 64;
        c += (u128)m * p[2] + t2; t1 = (uint64_t)c; c >>= 64;
        c += (u128)m * p[3] + t3; t2 = (uint64_t)c; c >>= 64;
        c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);
    }
    uint64_t t[4] = {t0, t1, t2, t3};
    if (t4 || limbs_geq(t, p)) limbs_sub(t, t, p);
    r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
}

#define DEFINE_FIELD(pfx, P, INV)                                                                   \
ORC_INLINE void pfx##_mul(fe_t *r, const fe_t *a, const fe_t *b) { mont_mul(r->l, a->l, b->l, P, INV); } \
ORC_INLINE void pfx##_sqr(fe_t *r, const fe_t *a) { mont_mul(r->l, a->l, a->l, P, INV); }           \
ORC_INLINE void pfx##_add(fe_t *r, const fe_t *a, const fe_t *b) {                                  \
    uint64_t t[4]; uint64_t c = limbs_add(t, a->l, b->l);                                           \
    if (c || limbs_geq(t, P)) limbs_sub(t, t, P);                                                   \
    memcpy(r->l, t, 32); }                                                                          \
ORC_INLINE void pfx##_sub(fe_t *r, const fe_t *a, const fe_t *b) {                                  \
    uint64_t t[4]; if (limbs_sub(t, a->l, b->l)) limbs_add(t, t, P);                                \
    memcpy(r->l, t, 32); }                                                                          \
ORC_INLINE int pfx##_is_zero(const fe_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; } \
ORC_INLINE int pfx##_eq(const fe_t *a, const fe_t *b) { return memcmp(a->l, b->l, 32) == 0; }       \
ORC_INLINE void pfx##_neg(fe_t *r, const fe_t *a) {                                                 \
    if (pfx##_is_zero(a)) { *r = *a; return; } uint64_t t[4]; limbs_sub(t, P, a->l); memcpy(r->l, t, 32); } \
ORC_INLINE void pfx##_dbl(fe_t *r, const fe_t *a) { pfx##_add(r, a, a); }                           \
/* canonical integer -> Montgomery: a * R^2 * R^-1 */                                               \
extern const fe_t pfx##_R2, pfx##_ONE;                                                              \
ORC_INLINE void pfx##_to_mont(fe_t *r, const fe_t *a) { pfx##_mul(r, a, &pfx##_R2); }               \
ORC_INLINE void pfx##_from_mont(fe_t *r, const fe_t *a) {                                           \
    fe_t one = {{1, 0, 0, 0}}; pfx##_mul(r, a, &one); }                                             \
ORC_INLINE void pfx##_pow(fe_t *r, const fe_t *a, const uint64_t e[4]) {                            \
    fe_t acc = pfx##_ONE, base = *a;                                                                \
    for (int i = 0; i < 256; i++) {                                                                 \
        if ((e[i >> 6] >> (i & 63)) & 1) pfx##_mul(&acc, &acc, &base);                              \
        pfx##_sqr(&base, &base); }                                                                  \
    *r = acc; }                                                                                     \
ORC_INLINE void pfx##_inv(fe_t *r, const fe_t *a) { /* Fermat: a^(p-2) */                           \
    uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]}; pfx##_pow(r, a, e); }                             \
ORC_INLINE void pfx##_set_u64(fe_t *r, uint64_t v) { fe_t t = {{v, 0, 0, 0}}; pfx##_to_mont(r, &t); }

DEFINE_FIELD(fr, FR_P, FR_INV)
DEFINE_FIELD(fq, FQ_P, FQ_INV)

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1) */
typedef struct { fe_t c0, c1; } fq2_t;
ORC_INLINE void fq2_add(fq2_t *r, const fq2_t *a, const fq2_t *b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
ORC_INLINE void fq2_sub(fq2_t *r, const fq2_t *a, const fq2_t *b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
ORC_INLINE void fq2_neg(fq2_t *r, const fq2_t *a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
ORC_INLINE void fq2_dbl(fq2_t *r, const fq2_t *a) { fq2_add(r, a, a); }
ORC_INLINE int fq2_is_zero(const fq2_t *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
ORC_INLINE int fq2_eq(const fq2_t *a, const fq2_t *b) { return fq_eq(&a->c0, &b->c0) && fq_eq(&a->c1, &b->c1); }
ORC_INLINE void fq2_mul(fq2_t *r, const fq2_t *a, const fq2_t *b) {   /* Karatsuba, 3 Fq mul */
    fe_t v0, v1, s, t;
    fq_mul(&v0, &a->c0, &b->c0); fq_mul(&v1, &a->c1, &b->c1);
    fq_add(&s, &a->c0, &a->c1); fq_add(&t, &b->c0, &b->c1);
    fq_mul(&s, &s, &t); fq_sub(&s, &s, &v0); fq_sub(&s, &s, &v1);
    fq_sub(&r->c0, &v0, &v1); r->c1 = s;
}
ORC_INLINE void fq2_sqr(fq2_t *r, const fq2_t *a) {                    /* (a0+a1)(a0-a1), 2 a0 a1 */
    fe_t s, d, p;
    fq_add(&s, &a->c0, &a->c1); fq_sub(&d, &a->c0, &a->c1); fq_mul(&p, &a->c0, &a->c1);
    fq_mul(&r->c0, &s, &d); fq_add(&r->c1, &p, &p);
}
ORC_INLINE void fq2_inv(fq2_t *r, const fq2_t *a) {
    fe_t n, t;
    fq_sqr(&n, &a->c0); fq_sqr(&t, &a->c1); fq_add(&n, &n, &t); fq_inv(&n, &n);
    fq_mul(&r->c0, &a->c0, &n); fq_mul(&t, &a->c1, &n); fq_neg(&r->c1, &t);
}
#endif
