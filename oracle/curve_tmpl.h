/*
 * oracle/curve_tmpl.h -- TEST INFRASTRUCTURE (CPU oracle).  Included twice by oracle.c:
 *   G1 over Fq  (y^2 = x^3 + 3)          -- libff alt_bn128_G1 [ABSENT from /root/reference]
 *   G2 over Fq2 (y^2 = x^3 + 3/(9+u))    -- libff alt_bn128_G2 [ABSENT]
 * Jacobian coordinates (X/Z^2, Y/Z^3), infinity = (Z == 0); a = 0 short Weierstrass formulas
 * (dbl-2009-l, madd-2007-bl, add-2007-bl from the Explicit-Formulas Database).
 * Affine points are {x, y}; the pair (0, 0) -- not on either curve -- encodes infinity.
 *
 * Required macros: FE (element type), FN(op) (field function name), PN(name) (curve function name),
 *                  JAC / AFF (point type names).
 */
typedef struct { FE X, Y, Z; } JAC;
typedef struct { FE x, y; } AFF;

ORC_INLINE int PN(aff_is_inf)(const AFF *p) { return FN(is_zero)(&p->x) && FN(is_zero)(&p->y); }
ORC_INLINE int PN(is_inf)(const JAC *p) { return FN(is_zero)(&p->Z); }
ORC_INLINE void PN(set_inf)(JAC *p) { memset(p, 0, sizeof(*p)); }

static void PN(from_aff)(JAC *r, const AFF *p, const FE *one) {
    if (PN(aff_is_inf)(p)) { PN(set_inf)(r); return; }
    r->X = p->x; r->Y = p->y; r->Z = *one;
}

static void PN(dbl)(JAC *r, const JAC *p) {
    if (PN(is_inf)(p)) { *r = *p; return; }
    FE A, B, C, D, E, F, t, Z3;
    FN(sqr)(&A, &p->X); FN(sqr)(&B, &p->Y); FN(sqr)(&C, &B);
    FN(add)(&t, &p->X, &B); FN(sqr)(&t, &t); FN(sub)(&t, &t, &A); FN(sub)(&t, &t, &C); FN(dbl)(&D, &t);
    FN(dbl)(&E, &A); FN(add)(&E, &E, &A);
    FN(sqr)(&F, &E);
    FN(mul)(&Z3, &p->Y, &p->Z); FN(dbl)(&Z3, &Z3);
    FN(dbl)(&t, &D); FN(sub)(&r->X, &F, &t);
    FN(sub)(&t, &D, &r->X); FN(mul)(&t, &E, &t);
    FN(dbl)(&C, &C); FN(dbl)(&C, &C); FN(dbl)(&C, &C);
    FN(sub)(&r->Y, &t, &C);
    r->Z = Z3;
}

/* r = p + q, q affine ("mixed addition", the *_with_mixed_addition paths of tcc:488-530) */
static void PN(madd)(JAC *r, const JAC *p, const AFF *q, const FE *one) {
    if (PN(aff_is_inf)(q)) { *r = *p; return; }
    if (PN(is_inf)(p)) { r->X = q->x; r->Y = q->y; r->Z = *one; return; }
    FE Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;
    FN(sqr)(&Z1Z1, &p->Z);
    FN(mul)(&U2, &q->x, &Z1Z1);
    FN(mul)(&S2, &q->y, &p->Z); FN(mul)(&S2, &S2, &Z1Z1);
    FN(sub)(&H, &U2, &p->X);
    FN(sub)(&rr, &S2, &p->Y);
    if (FN(is_zero)(&H)) {
        if (FN(is_zero)(&rr)) { PN(dbl)(r, p); return; }
        PN(set_inf)(r); return;
    }
    FN(dbl)(&rr, &rr);
    FN(sqr)(&HH, &H);
    FN(dbl)(&I, &HH); FN(dbl)(&I, &I);
    FN(mul)(&J, &H, &I);
    FN(mul)(&V, &p->X, &I);
    FN(sqr)(&X3, &rr); FN(sub)(&X3, &X3, &J); FN(dbl)(&t, &V); FN(sub)(&X3, &X3, &t);
    FN(sub)(&t, &V, &X3); FN(mul)(&Y3, &rr, &t); FN(mul)(&t, &p->Y, &J); FN(dbl)(&t, &t); FN(sub)(&Y3, &Y3, &t);
    FN(add)(&Z3, &p->Z, &H); FN(sqr)(&Z3, &Z3); FN(sub)(&Z3, &Z3, &Z1Z1); FN(sub)(&Z3, &Z3, &HH);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

static void PN(add)(JAC *r, const JAC *p, const JAC *q) {
    if (PN(is_inf)(p)) { *r = *q; return; }
    if (PN(is_inf)(q)) { *r = *p; return; }
    FE Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
    FN(sqr)(&Z1Z1, &p->Z); FN(sqr)(&Z2Z2, &q->Z);
    FN(mul)(&U1, &p->X, &Z2Z2); FN(mul)(&U2, &q->X, &Z1Z1);
    FN(mul)(&S1, &p->Y, &q->Z); FN(mul)(&S1, &S1, &Z2Z2);
    FN(mul)(&S2, &q->Y, &p->Z); FN(mul)(&S2, &S2, &Z1Z1);
    FN(sub)(&H, &U2, &U1);
    FN(sub)(&rr, &S2, &S1);
    if (FN(is_zero)(&H)) {
        if (FN(is_zero)(&rr)) { PN(dbl)(r, p); return; }
        PN(set_inf)(r); return;
    }
    FN(dbl)(&rr, &rr);
    FN(dbl)(&I, &H); FN(sqr)(&I, &I);
    FN(mul)(&J, &H, &I);
    FN(mul)(&V, &U1, &I);
    FN(sqr)(&X3, &rr); FN(sub)(&X3, &X3, &J); FN(dbl)(&t, &V); FN(sub)(&X3, &X3, &t);
    FN(sub)(&t, &V, &X3); FN(mul)(&Y3, &rr, &t); FN(mul)(&t, &S1, &J); FN(dbl)(&t, &t); FN(sub)(&Y3, &Y3, &t);
    FN(add)(&Z3, &p->Z, &q->Z); FN(sqr)(&Z3, &Z3); FN(sub)(&Z3, &Z3, &Z1Z1); FN(sub)(&Z3, &Z3, &Z2Z2);
    FN(mul)(&Z3, &Z3, &H);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

static void PN(neg_aff)(AFF *r, const AFF *p) { r->x = p->x; FN(neg)(&r->y, &p->y); }

/* to_affine_coordinates; infinity -> (0,0) in this oracle's affine encoding */
static void PN(to_aff)(AFF *r, const JAC *p) {
    if (PN(is_inf)(p)) { memset(r, 0, sizeof(*r)); return; }
    FE zi, zi2, zi3;
    FN(inv)(&zi, &p->Z); FN(sqr)(&zi2, &zi); FN(mul)(&zi3, &zi2, &zi);
    FN(mul)(&r->x, &p->X, &zi2); FN(mul)(&r->y, &p->Y, &zi3);
}

/* scalar given as canonical 256-bit integer (4 x u64 LE); plain double-and-add, MSB first */
static void PN(mul_scalar)(JAC *r, const JAC *p, const uint64_t k[4]) {
    JAC acc; PN(set_inf)(&acc);
    for (int i = 255; i >= 0; i--) {
        PN(dbl)(&acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) PN(add)(&acc, &acc, p);
    }
    *r = acc;
}

ORC_INLINE uint32_t PN(window)(const uint64_t k[4], unsigned bit, unsigned c) {
    if (bit >= 256) return 0;
    unsigned limb = bit >> 6, off = bit & 63;
    uint64_t v = k[limb] >> off;
    if (off + c > 64 && limb < 3) v |= k[limb + 1] << (64 - off);
    return (uint32_t)(v & ((1ULL << c) - 1));
}

/*
 * Sum_i scalar_i * base_i  -- libff::multi_exp<..., multi_exp_method_BDLO12> [ABSENT body],
 * call sites tcc:488-530.  Bucket (Pippenger) method over canonical scalars; scalars 0 are skipped
 * and scalars 1 take a single mixed addition (the *_with_mixed_addition partition, Appendix A.5).
 * `scalars` are canonical (already out of Montgomery form).  Windows are processed in parallel.
 */
static void PN(msm)(JAC *out, const AFF *bases, const fe_t *scalars, size_t n, const FE *one, unsigned c_override) {
    unsigned c = c_override;
    if (!c) { c = 1; while ((1ULL << (c + 3)) < n + 1 && c < 16) c++; if (c < 2) c = 2; }
    unsigned nwin = (254 + c - 1) / c;
    JAC *wsum = (JAC *)malloc(sizeof(JAC) * nwin);
    JAC ones; PN(set_inf)(&ones);
    #pragma omp parallel for schedule(dynamic, 1)
    for (int w = -1; w < (int)nwin; w++) {
        if (w < 0) {                                   /* the "scalar == 1" partition */
            JAC acc; PN(set_inf)(&acc);
            for (size_t i = 0; i < n; i++) {
                const uint64_t *k = scalars[i].l;
                if (k[0] == 1 && !(k[1] | k[2] | k[3])) PN(madd)(&acc, &acc, &bases[i], one);
            }
            ones = acc;
            continue;
        }
        size_t nb = ((size_t)1 << c) - 1;
        JAC *bk = (JAC *)calloc(nb, sizeof(JAC));
        for (size_t i = 0; i < n; i++) {
            const uint64_t *k = scalars[i].l;
            if (!(k[1] | k[2] | k[3]) && k[0] <= 1) continue;
            uint32_t d = PN(window)(k, (unsigned)w * c, c);
            if (d) PN(madd)(&bk[d - 1], &bk[d - 1], &bases[i], one);
        }
        JAC run, tot; PN(set_inf)(&run); PN(set_inf)(&tot);
        for (size_t b = nb; b-- > 0;) { PN(add)(&run, &run, &bk[b]); PN(add)(&tot, &tot, &run); }
        wsum[w] = tot;
        free(bk);
    }
    JAC acc; PN(set_inf)(&acc);
    for (int w = (int)nwin - 1; w >= 0; w--) {
        for (unsigned k = 0; k < c; k++) PN(dbl)(&acc, &acc);
        PN(add)(&acc, &acc, &wsum[w]);
    }
    PN(add)(&acc, &acc, &ones);
    free(wsum);
    *out = acc;
}

/* naive reference for the MSM itself: independent double-and-add per term */
static void PN(msm_naive)(JAC *out, const AFF *bases, const fe_t *scalars, size_t n, const FE *one) {
    JAC acc; PN(set_inf)(&acc);
    for (size_t i = 0; i < n; i++) {
        JAC p, t; PN(from_aff)(&p, &bases[i], one);
        PN(mul_scalar)(&t, &p, scalars[i].l);
        PN(add)(&acc, &acc, &t);
    }
    *out = acc;
}

/* fixed-base windowed batch exponentiation: out[i] = scalars[i] * g (keygen, tcc:358-411 batch_exp) */
static void PN(batch_mul)(AFF *out, const JAC *g, const fe_t *scalars, size_t n, const FE *one) {
    const unsigned c = 8, nwin = 32;
    static AFF *table = NULL;                                   /* table[w][d-1] = d * 2^(8w) * g; g is fixed per group */
    if (!table) {
        AFF *tb = (AFF *)malloc(sizeof(AFF) * nwin * 255);
        JAC base = *g;
        for (unsigned w = 0; w < nwin; w++) {
            JAC acc = base;
            for (unsigned d = 1; d <= 255; d++) {
                PN(to_aff)(&tb[w * 255 + d - 1], &acc);
                PN(add)(&acc, &acc, &base);
            }
            base = acc;                                         /* 256 * base */
        }
        table = tb;
    }
    #pragma omp parallel for schedule(static, 64)
    for (size_t i = 0; i < n; i++) {
        JAC acc; PN(set_inf)(&acc);
        for (unsigned w = 0; w < nwin; w++) {
            uint32_t d = PN(window)(scalars[i].l, w * c, c);
            if (d) PN(madd)(&acc, &acc, &table[w * 255 + d - 1], one);
        }
        PN(to_aff)(&out[i], &acc);
    }
}
