// tools/fp52.hpp -- MEASUREMENT ONLY (tools/mulbench.cpp, tools/fp52_check.cpp): the alternative multiplier substrate the round-3 review asked to be
// measured rather than priced -- Montgomery products over Fq on the FP64 pipe.  An element is 5 limbs of 52 bits held as doubles (exact integers),
// R = 2^260.  A 52 x 52-bit limb product comes out of two v_fma_f64 (Emmart / Luitjens / Weems, "Faster modular exponentiation using double
// precision floating point arithmetic on the GPU", ARITH 2018), in round-toward-zero mode:
//     hi = fma(a, b, 2^104)                = 2^104 + floor(ab / 2^52) 2^52        (ulp of [2^104, 2^105) is 2^52: the low half is cut off)
//     lo = fma(a, b, (2^104 + 2^52) - hi)  = 2^52 + (ab mod 2^52)                 (exact)
// so the raw IEEE bit patterns of hi and lo carry the two halves in their mantissas, and the column sums of a product-scanning multiplication are
// plain 64-bit integer additions of those patterns (the exponent fields add up to constants known at compile time and are pre-subtracted).
// Per limb product: 3 FP64 instructions + 2 64-bit integer additions, against 1 v_mad_u64_u32 + 1 v_addc per 32 x 32-bit product of the shipped form
// (fips_asm.hpp) -- 25 limb products instead of 64.  Nothing in ethsnarks_amd/ includes this file.
#pragma once
#include <stdint.h>
#include <string.h>
#ifndef ZK_HD
#define ZK_HD __host__ __device__ __forceinline__
#endif

namespace fp52 {

struct f52 { double l[5]; };

constexpr uint64_t M52 = (1ull << 52) - 1;
constexpr uint64_t OFF_HI = 0x4670000000000000ull;     // bits(2^104)
constexpr uint64_t OFF_LO = 0x4330000000000000ull;     // bits(2^52)
// q in 52-bit limbs, -q^-1 mod 2^52
constexpr uint64_t QL[5] = {0x8c16d87cfd47ull, 0x916871ca8d3c2ull, 0x181585d97816aull, 0xa029b85045b68ull, 0x30644e72e131ull};
constexpr uint64_t QINV = 0x20782e4866389ull;

static ZK_HD uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static ZK_HD double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
// the FP64 operations go through these two, so that neither contraction nor reassociation can touch them
static ZK_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// integer < 2^52 -> double, exactly: splice it into the mantissa of 2^52 and take 2^52 away
static ZK_HD double to_double(uint64_t v) { return from_bits(v | OFF_LO) - 4503599627370496.0; }

// (hi, lo) bit patterns of a * b, a and b integers in [0, 2^52) held as doubles; ROUND-TOWARD-ZERO mode
static ZK_HD void mulhl(double a, double b, uint64_t &hi, uint64_t &lo) {
    const double C1 = 20282409603651670423947251286016.0;                  // 2^104
    const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;   // 2^104 + 2^52 (representable: a multiple of 2^52)
    const double h = fma_(a, b, C1);
    const double s = C2 - h;
    const double l = fma_(a, b, s);
    hi = bits(h); lo = bits(l);
}

// number of (i, j) pairs with i + j == k, 0 <= i, j < 5
constexpr int pairs(int k) { return k < 0 || k > 8 ? 0 : (k < 5 ? k + 1 : 9 - k); }
// what column k starts from: minus the exponent patterns of every term that will ever land in it (NP a*b products and one m*q product)
constexpr uint64_t col_init(int k, int NP) {
    return 0ull - ((uint64_t)((NP + 1) * pairs(k)) * OFF_LO + (uint64_t)((NP + 1) * pairs(k - 1)) * OFF_HI);
}

// sum of NP products a[i] * b[i] with ONE Montgomery reduction: result < q (1 + 4 NP q / 2^260) < 2q for NP <= 4, limbs < 2^52
template <int NP>
static ZK_HD f52 mul_n(const f52 *const (&op)[2 * NP]) {
    uint64_t acc[10];
#pragma unroll
    for (int k = 0; k < 10; k++) acc[k] = col_init(k, NP);
#pragma unroll
    for (int t = 0; t < NP; t++)
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 5; j++) {
                uint64_t hi, lo;
                mulhl(op[2 * t]->l[i], op[2 * t + 1]->l[j], hi, lo);
                acc[i + j] += lo; acc[i + j + 1] += hi;
            }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        // column k is complete but for lo(m_k q_0), whose exponent pattern is already subtracted: only bits >= 52 are off
        const double tl = to_double(acc[k] & M52);
        uint64_t mh, ml;
        mulhl(tl, (double)QINV, mh, ml);
        const double m = from_bits(ml) - 4503599627370496.0;              // (t * -q^-1) mod 2^52
#pragma unroll
        for (int j = 0; j < 5; j++) {
            uint64_t hi, lo;
            mulhl(m, (double)QL[j], hi, lo);
            acc[k + j] += lo; acc[k + j + 1] += hi;
        }
        acc[k + 1] += acc[k] >> 52;                                        // the low 52 bits are zero now
    }
    f52 r;
#pragma unroll
    for (int k = 5; k < 9; k++) { r.l[k - 5] = to_double(acc[k] & M52); acc[k + 1] += acc[k] >> 52; }
    r.l[4] = to_double(acc[9]);
    return r;
}
static ZK_HD f52 mul(const f52 &a, const f52 &b) { const f52 *const op[2] = {&a, &b}; return mul_n<1>(op); }
static ZK_HD f52 mul2(const f52 &a, const f52 &b, const f52 &c, const f52 &d) { const f52 *const op[4] = {&a, &b, &c, &d}; return mul_n<2>(op); }

// limb-wise sum, carries propagated (operands < 2q: the sum < 4q < 2^256 fits the five limbs); no modular fold -- enough for the chains measured
static ZK_HD f52 add_nofold(const f52 &a, const f52 &b) {
    f52 r; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint64_t s = (bits(a.l[i] + 4503599627370496.0) & M52) + (bits(b.l[i] + 4503599627370496.0) & M52) + c;
        r.l[i] = to_double(i < 4 ? (s & M52) : s); c = s >> 52;
    }
    return r;
}

// 8 x u32 little-endian (value < 2^256) <-> five 52-bit limbs
static ZK_HD f52 from_u32(const uint32_t v[8]) {
    uint64_t w[4];
    for (int i = 0; i < 4; i++) w[i] = v[2 * i] | ((uint64_t)v[2 * i + 1] << 32);
    f52 r;
    r.l[0] = to_double(w[0] & M52);
    r.l[1] = to_double(((w[0] >> 52) | (w[1] << 12)) & M52);
    r.l[2] = to_double(((w[1] >> 40) | (w[2] << 24)) & M52);
    r.l[3] = to_double(((w[2] >> 28) | (w[3] << 36)) & M52);
    r.l[4] = to_double(w[3] >> 16);
    return r;
}
static ZK_HD void to_u32(const f52 &a, uint32_t v[8]) {
    uint64_t l[5];
    for (int i = 0; i < 5; i++) l[i] = bits(a.l[i] + 4503599627370496.0) & M52;       // (limbs < 2^52)
    uint64_t w[4];
    w[0] = l[0] | (l[1] << 52);
    w[1] = (l[1] >> 12) | (l[2] << 40);
    w[2] = (l[2] >> 24) | (l[3] << 28);
    w[3] = (l[3] >> 36) | (l[4] << 16);
    for (int i = 0; i < 4; i++) { v[2 * i] = (uint32_t)w[i]; v[2 * i + 1] = (uint32_t)(w[i] >> 32); }
}

}  // namespace fp52
