// bn254.hpp -- alt_bn128 field and curve arithmetic for the HIP backend (gfx950) and its host side.
//
// This is the from-scratch device equivalent of what the reference gets from libff
// (depends/libsnark/depends/libff [ABSENT in the checkout]): Fr, Fq, Fq2, G1, G2.  Used by
//   r1cs_gg_ppzksnark_zok_prover  (src/r1cs_gg_ppzksnark_zok/r1cs_gg_ppzksnark_zok.tcc:451-550).
//
// Representation: 8 x u32 little-endian limbs, Montgomery form with R = 2^256 -- bit-for-bit the
// 4 x u64 libff::Fp_model<4> memory image, so `pb.values` and the `.raw` key need no conversion.
// 32-bit limbs because the CDNA4 VALU multiplies 32x32 (v_mad_u64_u32 / v_mul_hi_u32); everything is
// fully unrolled so a field element lives in 8 VGPRs.  No MFMA: there is no dense contraction here.
//
// Curve points: affine {x, y} with (0,0) = infinity (neither curve contains it); accumulators use
// extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ; ZZ = 0 is infinity): mixed addition is
// 8M + 2S with no inversion, which is what bucket accumulation wants.
#pragma once
#include "common.hpp"
#include <stdint.h>
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
#include "fips_asm.hpp"
#endif

#ifndef ZK_HD
#define ZK_HD __host__ __device__ __forceinline__
#endif
#define ZK_HD_NOINLINE __host__ __device__ __attribute__((noinline))

namespace zk {

struct alignas(16) fe { uint32_t l[8]; };

struct FrParams {
    static ZK_HD constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return v[i];
    }
    static ZK_HD constexpr uint32_t one(int i) {   // R mod r
        constexpr uint32_t v[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    static ZK_HD constexpr uint32_t r2(int i) {    // R^2 mod r
        constexpr uint32_t v[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return v[i];
    }
    static constexpr uint32_t inv = 0xefffffffu;   // -r^-1 mod 2^32
    static constexpr bool is_fq = false;
};
struct FqParams {
    static ZK_HD constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return v[i];
    }
    static ZK_HD constexpr uint32_t one(int i) {   // R mod q
        constexpr uint32_t v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    static ZK_HD constexpr uint32_t r2(int i) {    // R^2 mod q
        constexpr uint32_t v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return v[i];
    }
    static constexpr uint32_t inv = 0xe4866389u;   // -q^-1 mod 2^32
    static constexpr bool is_fq = true;
};

// ------------------------------------------------------------------------------------------------
// prime field, fully reduced representatives in [0, p)
template <class P>
struct Field {
    typedef fe elem;

    static ZK_HD fe zero() { fe r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }
    static ZK_HD fe one() { fe r; for (int i = 0; i < 8; i++) r.l[i] = P::one(i); return r; }
    static ZK_HD bool is_zero(const fe &a) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) t |= a.l[i];
        return t == 0;
    }
    static ZK_HD bool eq(const fe &a, const fe &b) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) t |= a.l[i] ^ b.l[i];
        return t == 0;
    }
    // r = a - p if a >= p else a   (a < 2p)
    static ZK_HD fe reduce_once_c(const fe &a) {
        fe d; uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.l[i] - P::p(i) - br;
            d.l[i] = (uint32_t)t; br = (t >> 32) & 1;
        }
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = br ? a.l[i] : d.l[i];
        return r;
    }
    static ZK_HD fe add_c(const fe &a, const fe &b) {
        fe s; uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)a.l[i] + b.l[i]; s.l[i] = (uint32_t)c; c >>= 32; }
        return reduce_once_c(s);   // p < 2^254 so a + b < 2^255: no carry out of limb 7
    }
    static ZK_HD fe sub_c(const fe &a, const fe &b) {
        fe d; uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.l[i] - b.l[i] - br;
            d.l[i] = (uint32_t)t; br = (t >> 32) & 1;
        }
        uint32_t mask = br ? 0xffffffffu : 0u;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)d.l[i] + (P::p(i) & mask); d.l[i] = (uint32_t)c; c >>= 32; }
        return d;
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    // gfx950 forms: explicit VCC carry chains (hipcc lowers the portable 64-bit arithmetic to ~4x the instructions)
    static __device__ __forceinline__ fe reduce_once(const fe &a) {
        fe r = a;
        if constexpr (P::is_fq) fips::reduce8_fq(r.l); else fips::reduce8_fr(r.l);
        return r;
    }
    static __device__ __forceinline__ fe add(const fe &a, const fe &b) {
        fe s; fips::add8(s.l, a.l, b.l);
        if constexpr (P::is_fq) fips::reduce8_fq(s.l); else fips::reduce8_fr(s.l);
        return s;
    }
    static __device__ __forceinline__ fe sub(const fe &a, const fe &b) {
        fe d; uint32_t mask = fips::sub8(d.l, a.l, b.l);
        if constexpr (P::is_fq) fips::addp_masked8_fq(d.l, mask); else fips::addp_masked8_fr(d.l, mask);
        return d;
    }
#else
    static ZK_HD fe reduce_once(const fe &a) { return reduce_once_c(a); }
    static ZK_HD fe add(const fe &a, const fe &b) { return add_c(a, b); }
    static ZK_HD fe sub(const fe &a, const fe &b) { return sub_c(a, b); }
#endif
    static ZK_HD fe neg(const fe &a) { return is_zero(a) ? a : sub(zero(), a); }
    static ZK_HD fe dbl(const fe &a) { return add(a, a); }

    // Montgomery product a*b*R^-1 mod p.  Portable form: CIOS over 32-bit limbs (host code and the CPU
    // emulation harness); device form: mul_fips below.
    static ZK_HD fe mul_cios(const fe &a, const fe &b) {
        uint32_t t[9];
#pragma unroll
        for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t c = 0;
            const uint32_t bi = b.l[i];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                c = (uint64_t)a.l[j] * bi + t[j] + c;
                t[j] = (uint32_t)c; c >>= 32;
            }
            c += t[8];
            t[8] = (uint32_t)c;                       // t < 2^288 throughout: no 10th limb (p < 2^254)
            const uint32_t m = t[0] * P::inv;
            c = (uint64_t)m * P::p(0) + t[0];
            c >>= 32;
#pragma unroll
            for (int j = 1; j < 8; j++) {
                c = (uint64_t)m * P::p(j) + t[j] + c;
                t[j - 1] = (uint32_t)c; c >>= 32;
            }
            c += t[8];
            t[7] = (uint32_t)c; t[8] = (uint32_t)(c >> 32);
        }
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = t[i];
        return reduce_once_c(r);                      // result < 2p < 2^255, t[8] == 0
    }

#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    // gfx950 form: finely integrated product scanning.  Column k of a*b + m*p is summed into a 96-bit
    // accumulator (64-bit VGPR pair + 32-bit carry word): every partial product is ONE v_mad_u64_u32 whose
    // 64-bit addend is the accumulator itself, plus ONE v_addc_co_u32 that banks the carry-out the mad
    // leaves in an SGPR pair.  The kernels built on it are bound by VALU issue: measured on MI355X the product
    // costs the SUM of its instructions' issue times (v_mad_u64_u32 ~5.9 cycles per wave and SIMD, simple VALU
    // ~2.4, tools/mulbench.cpp), multiplies and carry adds do not overlap.  hipcc cannot use that carry-out from C (it builds zero-extended addend pairs
    // with v_mov / v_lshl_add_u64 instead: 565 VALU instructions per product against ~330 here).
    // The modulus limbs ride in SGPRs (wave-uniform constants).  VALU-written SGPR carry -> VALU carry-in is
    // hardware-interlocked on gfx9 (the same pattern hipcc emits through VCC), so no s_nop is needed.
    // (hi:lo) += sum of the a*b products of column i, then of its m*p products: one asm statement each
    template <int N, int J0, int I>
    static __device__ __forceinline__ void col_ab(uint64_t &lo, uint32_t &hi, const fe &a, const fe &b) {
#define ZK_AB(k) a.l[J0 + k], b.l[I - J0 - k]
        if constexpr (N == 1) fips::mac1_vv(lo, hi, ZK_AB(0));
        if constexpr (N == 2) fips::mac2_vv(lo, hi, ZK_AB(0), ZK_AB(1));
        if constexpr (N == 3) fips::mac3_vv(lo, hi, ZK_AB(0), ZK_AB(1), ZK_AB(2));
        if constexpr (N == 4) fips::mac4_vv(lo, hi, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3));
        if constexpr (N == 5) fips::mac5_vv(lo, hi, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_AB(4));
        if constexpr (N == 6) fips::mac6_vv(lo, hi, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_AB(4), ZK_AB(5));
        if constexpr (N == 7) fips::mac7_vv(lo, hi, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_AB(4), ZK_AB(5), ZK_AB(6));
        if constexpr (N == 8) fips::mac8_vv(lo, hi, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_AB(4), ZK_AB(5), ZK_AB(6), ZK_AB(7));
#undef ZK_AB
    }
    template <int N, int J0, int I>
    static __device__ __forceinline__ void col_mp(uint64_t &lo, uint32_t &hi, const uint32_t (&m)[8]) {
#define ZK_MP(k) m[J0 + k], P::p(I - J0 - k)
        if constexpr (N == 1) fips::mac1_vs(lo, hi, ZK_MP(0));
        if constexpr (N == 2) fips::mac2_vs(lo, hi, ZK_MP(0), ZK_MP(1));
        if constexpr (N == 3) fips::mac3_vs(lo, hi, ZK_MP(0), ZK_MP(1), ZK_MP(2));
        if constexpr (N == 4) fips::mac4_vs(lo, hi, ZK_MP(0), ZK_MP(1), ZK_MP(2), ZK_MP(3));
        if constexpr (N == 5) fips::mac5_vs(lo, hi, ZK_MP(0), ZK_MP(1), ZK_MP(2), ZK_MP(3), ZK_MP(4));
        if constexpr (N == 6) fips::mac6_vs(lo, hi, ZK_MP(0), ZK_MP(1), ZK_MP(2), ZK_MP(3), ZK_MP(4), ZK_MP(5));
        if constexpr (N == 7) fips::mac7_vs(lo, hi, ZK_MP(0), ZK_MP(1), ZK_MP(2), ZK_MP(3), ZK_MP(4), ZK_MP(5), ZK_MP(6));
        if constexpr (N == 8) fips::mac8_vs(lo, hi, ZK_MP(0), ZK_MP(1), ZK_MP(2), ZK_MP(3), ZK_MP(4), ZK_MP(5), ZK_MP(6), ZK_MP(7));
#undef ZK_MP
    }
    // Column scheme.  The column accumulator is (hi:lo) = a 64-bit VGPR pair + a 32-bit carry word.  The FIRST multiply-add
    // of a column takes the previous column's upper 64 bits {lo.hi, hi} as its addend and SETS the carry word
    // (fips::macN_*_set: write-only, early-clobber outputs), so moving on to the next column is one v_mov (lo.hi into the
    // even register of the addend pair; gfx90a+ wants 64-bit operands even-aligned) instead of three (shift the pair, clear
    // the carry word).  NP = number of a*b products summed before the single reduction (1: product, 2 / 4: dot products).
    template <int N, int J0, int I>
    static __device__ __forceinline__ void col_ab_set(uint64_t &lo, uint32_t &hi, uint64_t ad, const fe &a, const fe &b) {
#define ZK_AB(k) a.l[J0 + k], b.l[I - J0 - k]
        if constexpr (N == 1) fips::mac1_vv_set(lo, hi, ad, ZK_AB(0));
        if constexpr (N == 2) fips::mac2_vv_set(lo, hi, ad, ZK_AB(0), ZK_AB(1));
        if constexpr (N == 3) fips::mac3_vv_set(lo, hi, ad, ZK_AB(0), ZK_AB(1), ZK_AB(2));
        if constexpr (N == 4) fips::mac4_vv_set(lo, hi, ad, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3));
        if constexpr (N == 5) fips::mac5_vv_set(lo, hi, ad, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_AB(4));
        if constexpr (N == 6) fips::mac6_vv_set(lo, hi, ad, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_AB(4), ZK_AB(5));
        if constexpr (N == 7) fips::mac7_vv_set(lo, hi, ad, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_AB(4), ZK_AB(5), ZK_AB(6));
        if constexpr (N == 8) fips::mac8_vv_set(lo, hi, ad, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_AB(4), ZK_AB(5), ZK_AB(6), ZK_AB(7));
#undef ZK_AB
    }
    // m_i = t_i * (-p^-1) mod 2^32 through the 64-bit multiplier: v_mad_u64_u32 issues faster than v_mul_lo_u32 on gfx950
    static __device__ __forceinline__ uint32_t mont_m(uint32_t t0) { return fips::mullo_vs(t0, P::inv); }
    static __device__ __forceinline__ uint64_t next_addend(uint64_t lo, uint32_t hi) { return (lo >> 32) | ((uint64_t)hi << 32); }
    template <int NP, bool REDUCE>
    static __device__ __forceinline__ fe mul_fips_n(const fe *const (&op)[2 * NP]) {
        uint64_t lo, ad; uint32_t hi;
        uint32_t m[8], t[8];
        // column 0: one product per pair, no addend
        fips::mul1_vv(lo, op[0]->l[0], op[1]->l[0]);
        if constexpr (NP >= 2) fips::mac1_vv_sethi(lo, hi, op[2]->l[0], op[3]->l[0]);
        if constexpr (NP >= 4) { fips::mac1_vv(lo, hi, op[4]->l[0], op[5]->l[0]); fips::mac1_vv(lo, hi, op[6]->l[0], op[7]->l[0]); }
        m[0] = mont_m((uint32_t)lo);
        if constexpr (NP == 1) fips::mac1_vs_sethi(lo, hi, m[0], P::p(0)); else fips::mac1_vs(lo, hi, m[0], P::p(0));
        ad = next_addend(lo, hi);
#define ZK_LOW(I) \
        col_ab_set<I + 1, 0, I>(lo, hi, ad, *op[0], *op[1]);   /* a_0 b_I + ... + a_I b_0 */ \
        if constexpr (NP >= 2) col_ab<I + 1, 0, I>(lo, hi, *op[2], *op[3]); \
        if constexpr (NP >= 4) { col_ab<I + 1, 0, I>(lo, hi, *op[4], *op[5]); col_ab<I + 1, 0, I>(lo, hi, *op[6], *op[7]); } \
        col_mp<I, 0, I>(lo, hi, m);                            /* m_0 p_I + ... + m_{I-1} p_1 */ \
        m[I] = mont_m((uint32_t)lo); \
        fips::mac1_vs(lo, hi, m[I], P::p(0));                  /* low word of the column is now 0 */ \
        ad = next_addend(lo, hi);
#define ZK_HIGH(I) \
        col_ab_set<15 - I, I - 7, I>(lo, hi, ad, *op[0], *op[1]);   /* a_{I-7} b_7 + ... + a_7 b_{I-7} */ \
        if constexpr (NP >= 2) col_ab<15 - I, I - 7, I>(lo, hi, *op[2], *op[3]); \
        if constexpr (NP >= 4) { col_ab<15 - I, I - 7, I>(lo, hi, *op[4], *op[5]); col_ab<15 - I, I - 7, I>(lo, hi, *op[6], *op[7]); } \
        col_mp<15 - I, I - 7, I>(lo, hi, m); \
        t[I - 8] = (uint32_t)lo; \
        ad = next_addend(lo, hi);
        ZK_LOW(1) ZK_LOW(2) ZK_LOW(3) ZK_LOW(4) ZK_LOW(5) ZK_LOW(6) ZK_LOW(7)
        ZK_HIGH(8) ZK_HIGH(9) ZK_HIGH(10) ZK_HIGH(11) ZK_HIGH(12) ZK_HIGH(13) ZK_HIGH(14)
#undef ZK_LOW
#undef ZK_HIGH
        t[7] = (uint32_t)ad;                               // column 15 is empty: just the carry of column 14
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = t[i];
        if constexpr (NP == 1) { if constexpr (REDUCE) return reduce_once(r); else return r; }   // a, b < 2p  =>  result < 2p (p < 2^254): see "loose domain" below
        if constexpr (NP >= 2) { if constexpr (P::is_fq) fips::reduce8_fq2(r.l); else fips::reduce8_fr2(r.l); }    // (8p^2 + Rp)/R < 2.51p: one fold by 2p
        if constexpr (NP >= 4) { if constexpr (P::is_fq) fips::reduce8_fq2(r.l); else fips::reduce8_fr2(r.l); }    // (16p^2 + Rp)/R < 4.03p: two folds
        return r;
    }
    template <bool REDUCE>
    static __device__ __forceinline__ fe mul_fips(const fe &a, const fe &b) { const fe *const op[2] = {&a, &b}; return mul_fips_n<1, REDUCE>(op); }
    static __device__ __forceinline__ fe mul(const fe &a, const fe &b) { return mul_fips<true>(a, b); }
    // a*b + c*d with ONE Montgomery reduction: both products are summed into the same columns before the m*p terms
    // (2 x 64 + 72 multiplies instead of 2 x 136, and no separate modular addition).  Loose inputs (< 2p) give
    // (8p^2 + Rp)/R < 2.51p, so one fold by 2p brings the result back into [0, 2p).  This is what an Fq2 product is
    // made of: c0 = a0 b0 + a1 (-b1), c1 = a0 b1 + a1 b0.
    static __device__ __forceinline__ fe lmul2(const fe &a, const fe &b, const fe &c, const fe &d) {
        const fe *const op[4] = {&a, &b, &c, &d}; return mul_fips_n<2, false>(op);
    }
    // a*b + c*d + e*f + g*h with one reduction: (16p^2 + Rp)/R < 4.03p, two folds by 2p (an Fq2 dot product of two terms)
    static __device__ __forceinline__ fe lmul4(const fe &a, const fe &b, const fe &c, const fe &d, const fe &e, const fe &f, const fe &g, const fe &h) {
        const fe *const op[8] = {&a, &b, &c, &d, &e, &f, &g, &h}; return mul_fips_n<4, false>(op);
    }
    // ---- dual forms: TWO independent products advance together, instruction by instruction, so that the serial tail of every
    // column of one (last multiply-add -> m_i -> m_i p_0 -> next column's addend) fills with the other's multiply-adds
    // (fips_asm.hpp, *_x2).  Measured (tools/mulbench.cpp, profiles/r03_dual_issue.txt): a chain of single products gains 14 % at
    // 2 waves/SIMD, 9 % at 4 (140 G/s = the issue bound); two-term dot products -- what Fq2 is made of -- gain nothing, their
    // columns are long enough.  Used by the G1 mixed addition (Curve::madd, PAIRS), which pays for the second set of m / t words
    // with one wave per SIMD (3 instead of 4: ZK_G1_WPS).
    template <int N, int J0, int I>
    static __device__ __forceinline__ void col_ab_x2(uint64_t &l0, uint32_t &h0, uint64_t &l1, uint32_t &h1, const fe &a, const fe &b, const fe &c, const fe &d) {
        if constexpr (N > 4) { col_ab_x2<4, J0, I>(l0, h0, l1, h1, a, b, c, d); col_ab_x2<N - 4, J0 + 4, I>(l0, h0, l1, h1, a, b, c, d); }
        else {
#define ZK_AB(k) a.l[J0 + k], b.l[I - J0 - k]
#define ZK_CD(k) c.l[J0 + k], d.l[I - J0 - k]
            if constexpr (N == 1) fips::mac1_vv_x2(l0, h0, l1, h1, ZK_AB(0), ZK_CD(0));
            if constexpr (N == 2) fips::mac2_vv_x2(l0, h0, l1, h1, ZK_AB(0), ZK_AB(1), ZK_CD(0), ZK_CD(1));
            if constexpr (N == 3) fips::mac3_vv_x2(l0, h0, l1, h1, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_CD(0), ZK_CD(1), ZK_CD(2));
            if constexpr (N == 4) fips::mac4_vv_x2(l0, h0, l1, h1, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_CD(0), ZK_CD(1), ZK_CD(2), ZK_CD(3));
        }
    }
    template <int N, int J0, int I>
    static __device__ __forceinline__ void col_ab_set_x2(uint64_t &l0, uint32_t &h0, uint64_t &l1, uint32_t &h1, uint64_t ad0, uint64_t ad1, const fe &a, const fe &b, const fe &c, const fe &d) {
        if constexpr (N > 4) { col_ab_set_x2<4, J0, I>(l0, h0, l1, h1, ad0, ad1, a, b, c, d); col_ab_x2<N - 4, J0 + 4, I>(l0, h0, l1, h1, a, b, c, d); }
        else {
            if constexpr (N == 1) fips::mac1_vv_set_x2(l0, h0, l1, h1, ad0, ad1, ZK_AB(0), ZK_CD(0));
            if constexpr (N == 2) fips::mac2_vv_set_x2(l0, h0, l1, h1, ad0, ad1, ZK_AB(0), ZK_AB(1), ZK_CD(0), ZK_CD(1));
            if constexpr (N == 3) fips::mac3_vv_set_x2(l0, h0, l1, h1, ad0, ad1, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_CD(0), ZK_CD(1), ZK_CD(2));
            if constexpr (N == 4) fips::mac4_vv_set_x2(l0, h0, l1, h1, ad0, ad1, ZK_AB(0), ZK_AB(1), ZK_AB(2), ZK_AB(3), ZK_CD(0), ZK_CD(1), ZK_CD(2), ZK_CD(3));
#undef ZK_AB
#undef ZK_CD
        }
    }
    template <int N, int J0, int I>
    static __device__ __forceinline__ void col_mp_x2(uint64_t &l0, uint32_t &h0, uint64_t &l1, uint32_t &h1, const uint32_t (&m0)[8], const uint32_t (&m1)[8]) {
        if constexpr (N > 4) { col_mp_x2<4, J0, I>(l0, h0, l1, h1, m0, m1); col_mp_x2<N - 4, J0 + 4, I>(l0, h0, l1, h1, m0, m1); }
        else {
#define ZK_P(k) P::p(I - J0 - k)
            if constexpr (N == 1) fips::mac1_vs_x2(l0, h0, l1, h1, m0[J0], m1[J0], ZK_P(0));
            if constexpr (N == 2) fips::mac2_vs_x2(l0, h0, l1, h1, m0[J0], m0[J0 + 1], m1[J0], m1[J0 + 1], ZK_P(0), ZK_P(1));
            if constexpr (N == 3) fips::mac3_vs_x2(l0, h0, l1, h1, m0[J0], m0[J0 + 1], m0[J0 + 2], m1[J0], m1[J0 + 1], m1[J0 + 2], ZK_P(0), ZK_P(1), ZK_P(2));
            if constexpr (N == 4) fips::mac4_vs_x2(l0, h0, l1, h1, m0[J0], m0[J0 + 1], m0[J0 + 2], m0[J0 + 3], m1[J0], m1[J0 + 1], m1[J0 + 2], m1[J0 + 3], ZK_P(0), ZK_P(1), ZK_P(2), ZK_P(3));
#undef ZK_P
        }
    }
    // (r0, r1) = (sum of NP products of opA, sum of NP products of opB), each with one Montgomery reduction, bounds as mul_fips_n
    template <int NP>
    static __device__ __forceinline__ void mul_fips_n_x2(const fe *const (&A)[2 * NP], const fe *const (&B)[2 * NP], fe &r0, fe &r1) {
        uint64_t l0, l1, ad0, ad1; uint32_t h0, h1;
        uint32_t m0[8], m1[8], t0[8], t1[8];
        fips::mul1_vv_x2(l0, l1, A[0]->l[0], A[1]->l[0], B[0]->l[0], B[1]->l[0]);
        if constexpr (NP >= 2) fips::mac1_vv_sethi_x2(l0, h0, l1, h1, A[2]->l[0], A[3]->l[0], B[2]->l[0], B[3]->l[0]);
        if constexpr (NP >= 4) { fips::mac1_vv_x2(l0, h0, l1, h1, A[4]->l[0], A[5]->l[0], B[4]->l[0], B[5]->l[0]); fips::mac1_vv_x2(l0, h0, l1, h1, A[6]->l[0], A[7]->l[0], B[6]->l[0], B[7]->l[0]); }
        fips::mullo_vs_x2(m0[0], m1[0], (uint32_t)l0, (uint32_t)l1, P::inv);
        if constexpr (NP == 1) fips::mac1_vs_sethi_x2(l0, h0, l1, h1, m0[0], m1[0], P::p(0)); else fips::mac1_vs_x2(l0, h0, l1, h1, m0[0], m1[0], P::p(0));
        ad0 = next_addend(l0, h0); ad1 = next_addend(l1, h1);
#define ZK_LOW(I) \
        col_ab_set_x2<I + 1, 0, I>(l0, h0, l1, h1, ad0, ad1, *A[0], *A[1], *B[0], *B[1]); \
        if constexpr (NP >= 2) col_ab_x2<I + 1, 0, I>(l0, h0, l1, h1, *A[2], *A[3], *B[2], *B[3]); \
        if constexpr (NP >= 4) { col_ab_x2<I + 1, 0, I>(l0, h0, l1, h1, *A[4], *A[5], *B[4], *B[5]); col_ab_x2<I + 1, 0, I>(l0, h0, l1, h1, *A[6], *A[7], *B[6], *B[7]); } \
        col_mp_x2<I, 0, I>(l0, h0, l1, h1, m0, m1); \
        fips::mullo_vs_x2(m0[I], m1[I], (uint32_t)l0, (uint32_t)l1, P::inv); \
        fips::mac1_vs_x2(l0, h0, l1, h1, m0[I], m1[I], P::p(0)); \
        ad0 = next_addend(l0, h0); ad1 = next_addend(l1, h1);
#define ZK_HIGH(I) \
        col_ab_set_x2<15 - I, I - 7, I>(l0, h0, l1, h1, ad0, ad1, *A[0], *A[1], *B[0], *B[1]); \
        if constexpr (NP >= 2) col_ab_x2<15 - I, I - 7, I>(l0, h0, l1, h1, *A[2], *A[3], *B[2], *B[3]); \
        if constexpr (NP >= 4) { col_ab_x2<15 - I, I - 7, I>(l0, h0, l1, h1, *A[4], *A[5], *B[4], *B[5]); col_ab_x2<15 - I, I - 7, I>(l0, h0, l1, h1, *A[6], *A[7], *B[6], *B[7]); } \
        col_mp_x2<15 - I, I - 7, I>(l0, h0, l1, h1, m0, m1); \
        t0[I - 8] = (uint32_t)l0; t1[I - 8] = (uint32_t)l1; \
        ad0 = next_addend(l0, h0); ad1 = next_addend(l1, h1);
        ZK_LOW(1) ZK_LOW(2) ZK_LOW(3) ZK_LOW(4) ZK_LOW(5) ZK_LOW(6) ZK_LOW(7)
        ZK_HIGH(8) ZK_HIGH(9) ZK_HIGH(10) ZK_HIGH(11) ZK_HIGH(12) ZK_HIGH(13) ZK_HIGH(14)
#undef ZK_LOW
#undef ZK_HIGH
        t0[7] = (uint32_t)ad0; t1[7] = (uint32_t)ad1;
#pragma unroll
        for (int i = 0; i < 8; i++) { r0.l[i] = t0[i]; r1.l[i] = t1[i]; }
        if constexpr (NP >= 2) { if constexpr (P::is_fq) { fips::reduce8_fq2(r0.l); fips::reduce8_fq2(r1.l); } else { fips::reduce8_fr2(r0.l); fips::reduce8_fr2(r1.l); } }
        if constexpr (NP >= 4) { if constexpr (P::is_fq) { fips::reduce8_fq2(r0.l); fips::reduce8_fq2(r1.l); } else { fips::reduce8_fr2(r0.l); fips::reduce8_fr2(r1.l); } }
    }
    // loose-domain pairs: (a b, c d); (a b + c d, e f + g h); and the four-term form
    static __device__ __forceinline__ void lmul_x2(const fe &a, const fe &b, const fe &c, const fe &d, fe &r0, fe &r1) {
        const fe *const A[2] = {&a, &b}; const fe *const B[2] = {&c, &d}; mul_fips_n_x2<1>(A, B, r0, r1);
    }
    static __device__ __forceinline__ void lmul2_x2(const fe &a, const fe &b, const fe &c, const fe &d, const fe &e, const fe &f, const fe &g, const fe &h, fe &r0, fe &r1) {
        const fe *const A[4] = {&a, &b, &c, &d}; const fe *const B[4] = {&e, &f, &g, &h}; mul_fips_n_x2<2>(A, B, r0, r1);   // (tools/mulbench.cpp measures it)
    }
#else
    static ZK_HD fe mul(const fe &a, const fe &b) { return mul_cios(a, b); }
#endif
    static ZK_HD fe sqr(const fe &a) { return mul(a, a); }

    // ---- "loose" domain [0, 2p): what the curve formulas compute in on the device.  With p < 2^254 = R/4 the
    // Montgomery product of two values < 2p is again < 2p WITHOUT the final conditional subtraction
    // ((4p^2 + Rp)/R < 2p), sums stay below 4p < 2^256, so only add/sub fold back (by 2p).  Zero has two
    // representatives (0 and p).  Strict operations accept loose inputs where noted (mul does; add/sub do not).
    // Host code and the CPU emulation map the loose names onto the strict operations (a subset of the domain).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    static __device__ __forceinline__ fe lmul(const fe &a, const fe &b) { return mul_fips<false>(a, b); }
    static __device__ __forceinline__ fe ladd(const fe &a, const fe &b) {
        fe s; fips::add8(s.l, a.l, b.l);
        if constexpr (P::is_fq) fips::reduce8_fq2(s.l); else fips::reduce8_fr2(s.l);
        return s;
    }
    static __device__ __forceinline__ fe lsub(const fe &a, const fe &b) {
        fe d; uint32_t mask = fips::sub8(d.l, a.l, b.l);
        if constexpr (P::is_fq) fips::addp_masked8_fq2(d.l, mask); else fips::addp_masked8_fr2(d.l, mask);
        return d;
    }
    static __device__ __forceinline__ bool lis_zero(const fe &a) {
        uint32_t z = 0, e = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { z |= a.l[i]; e |= a.l[i] ^ P::p(i); }
        return z == 0 || e == 0;
    }
#else
    static ZK_HD fe lmul(const fe &a, const fe &b) { return mul(a, b); }
    static ZK_HD fe lmul2(const fe &a, const fe &b, const fe &c, const fe &d) { return add(mul(a, b), mul(c, d)); }
    static ZK_HD fe lmul4(const fe &a, const fe &b, const fe &c, const fe &d, const fe &e, const fe &f, const fe &g, const fe &h) {
        return add(add(mul(a, b), mul(c, d)), add(mul(e, f), mul(g, h)));
    }
    static ZK_HD fe ladd(const fe &a, const fe &b) { return add(a, b); }
    static ZK_HD fe lsub(const fe &a, const fe &b) { return sub(a, b); }
    static ZK_HD bool lis_zero(const fe &a) { return is_zero(a); }
#endif
#if !(defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM))
    static ZK_HD void lmul_x2(const fe &a, const fe &b, const fe &c, const fe &d, fe &r0, fe &r1) { r0 = lmul(a, b); r1 = lmul(c, d); }
#endif
    // two independent products of the G1 formulas as one call (dual issue on the device; PAIRS: whether the formulas use it)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM) && !defined(ZK_NO_X2) && !defined(ZK_G1_NO_X2)
    static constexpr bool PAIRS = true;
#else
    static constexpr bool PAIRS = false;
#endif
    static ZK_HD void lmul_pair(const fe &a, const fe &b, const fe &c, const fe &d, fe &r0, fe &r1) { lmul_x2(a, b, c, d, r0, r1); }
    static ZK_HD void lsqr_pair(const fe &a, const fe &b, fe &r0, fe &r1) { lmul_pair(a, a, b, b, r0, r1); }
    static ZK_HD fe lsqr(const fe &a) { return lmul(a, a); }
    static ZK_HD fe ldbl(const fe &a) { return ladd(a, a); }
    static ZK_HD fe lneg(const fe &a) { return lsub(zero(), a); }
    // -a as an OPERAND OF A PRODUCT: 2p - a in (0, 2p], eight subtractions and no fold.  2p itself is outside the loose domain but
    // the product bounds hold for operands <= 2p ((4p^2 + Rp)/R < 2p, (8p^2 + Rp)/R < 2.51p, (16p^2 + Rp)/R < 4.03p), so the value
    // may feed lmul / lmul2 / lmul4 and nothing else (never compared, stored or added)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    static __device__ __forceinline__ fe lneg_op(const fe &a) {
        fe pp, d;
#pragma unroll
        for (int i = 0; i < 8; i++) pp.l[i] = (P::p(i) << 1) | (i ? P::p(i - 1) >> 31 : 0u);
        fips::sub8(d.l, pp.l, a.l);
        return d;
    }
#else
    static ZK_HD fe lneg_op(const fe &a) { return lneg(a); }
#endif
    // the negated Y operand of the curve formulas' dot products: the G1 kernels sit at their register limit and keep the folded
    // form (the unfolded one costs them a spill), Fq2 takes the cheap one
    static ZK_HD fe lneg_yop(const fe &a) { return lneg(a); }
    // quad helpers (Curve::*_q): lane ql of a 4-lane group picks its operand; qbcast<S> hands lane S's value to all four
    static ZK_HD fe qsel(uint32_t ql, const fe &a, const fe &b, const fe &c, const fe &d) {
        fe r; const bool odd = ql & 1u, hi = ql & 2u;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t x = odd ? b.l[i] : a.l[i], y = odd ? d.l[i] : c.l[i]; r.l[i] = hi ? y : x; }
        return r;
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    template <int S> static __device__ __forceinline__ fe qbcast(const fe &v) {
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], S * 0x55, 0xf, 0xf, false);   // quad_perm:[S,S,S,S]
        return r;
    }
#endif
    static ZK_HD fe canon(const fe &a) { return reduce_once(a); }       // loose -> [0, p)

    static ZK_HD fe to_mont(const fe &a) { fe r2; for (int i = 0; i < 8; i++) r2.l[i] = P::r2(i); return mul(a, r2); }
    static ZK_HD fe from_mont(const fe &a) { fe o = zero(); o.l[0] = 1; return mul(a, o); }

    static ZK_HD fe pow(const fe &a, const uint32_t e[8]) {
        fe acc = one(), base = a;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
    static ZK_HD fe pow_u64(const fe &a, uint64_t e) {
        fe acc = one(), base = a;
        while (e) { if (e & 1) acc = mul(acc, base); base = sqr(base); e >>= 1; }
        return acc;
    }
    static ZK_HD fe inv(const fe &a) {                // Fermat a^(p-2); host-side and rare device use only
        uint32_t e[8];
        for (int i = 0; i < 8; i++) e[i] = P::p(i);
        e[0] -= 2;                                    // p(0) >= 2 for both moduli, no borrow
        return pow(a, e);
    }
    static ZK_HD fe from_u64(uint64_t v) { fe t = zero(); t.l[0] = (uint32_t)v; t.l[1] = (uint32_t)(v >> 32); return to_mont(t); }
};

typedef Field<FrParams> Fr;
typedef Field<FqParams> Fq;

// ------------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2 + 1)
struct alignas(16) fe2 { fe c0, c1; };

struct Fq2 {
    typedef fe2 elem;
    static ZK_HD fe2 zero() { fe2 r; r.c0 = Fq::zero(); r.c1 = Fq::zero(); return r; }
    static ZK_HD fe2 one() { fe2 r; r.c0 = Fq::one(); r.c1 = Fq::zero(); return r; }
    static ZK_HD bool is_zero(const fe2 &a) { return Fq::is_zero(a.c0) && Fq::is_zero(a.c1); }
    static ZK_HD bool eq(const fe2 &a, const fe2 &b) { return Fq::eq(a.c0, b.c0) && Fq::eq(a.c1, b.c1); }
    static ZK_HD fe2 add(const fe2 &a, const fe2 &b) { fe2 r; r.c0 = Fq::add(a.c0, b.c0); r.c1 = Fq::add(a.c1, b.c1); return r; }
    static ZK_HD fe2 sub(const fe2 &a, const fe2 &b) { fe2 r; r.c0 = Fq::sub(a.c0, b.c0); r.c1 = Fq::sub(a.c1, b.c1); return r; }
    static ZK_HD fe2 neg(const fe2 &a) { fe2 r; r.c0 = Fq::neg(a.c0); r.c1 = Fq::neg(a.c1); return r; }
    static ZK_HD fe2 dbl(const fe2 &a) { return add(a, a); }
    static ZK_HD fe2 mul(const fe2 &a, const fe2 &b) {       // Karatsuba: 3 Fq products
        fe v0 = Fq::mul(a.c0, b.c0), v1 = Fq::mul(a.c1, b.c1);
        fe s = Fq::mul(Fq::add(a.c0, a.c1), Fq::add(b.c0, b.c1));
        fe2 r; r.c0 = Fq::sub(v0, v1); r.c1 = Fq::sub(Fq::sub(s, v0), v1); return r;
    }
    static ZK_HD fe2 sqr(const fe2 &a) {                      // complex squaring: 2 Fq products
        fe p = Fq::mul(a.c0, a.c1);
        fe2 r; r.c0 = Fq::mul(Fq::add(a.c0, a.c1), Fq::sub(a.c0, a.c1)); r.c1 = Fq::add(p, p); return r;
    }
    // loose-domain forms (see Field): componentwise
    static ZK_HD bool lis_zero(const fe2 &a) { return Fq::lis_zero(a.c0) && Fq::lis_zero(a.c1); }
    static ZK_HD fe2 ladd(const fe2 &a, const fe2 &b) { fe2 r; r.c0 = Fq::ladd(a.c0, b.c0); r.c1 = Fq::ladd(a.c1, b.c1); return r; }
    static ZK_HD fe2 lsub(const fe2 &a, const fe2 &b) { fe2 r; r.c0 = Fq::lsub(a.c0, b.c0); r.c1 = Fq::lsub(a.c1, b.c1); return r; }
    static ZK_HD fe2 lneg(const fe2 &a) { fe2 r; r.c0 = Fq::lneg(a.c0); r.c1 = Fq::lneg(a.c1); return r; }
    static ZK_HD fe2 lneg_op(const fe2 &a) { fe2 r; r.c0 = Fq::lneg_op(a.c0); r.c1 = Fq::lneg_op(a.c1); return r; }   // product operand only (Field::lneg_op)
    static ZK_HD fe2 lneg_yop(const fe2 &a) { return lneg_op(a); }
    static ZK_HD fe2 ldbl(const fe2 &a) { return ladd(a, a); }
    // schoolbook with one reduction per component (Field::lmul2): 4 x 64 + 2 x 72 multiplies and two folds, against
    // Karatsuba's 3 x 136 multiplies plus five modular additions -- fewer VALU issue slots on gfx950
    static ZK_HD fe2 lmul(const fe2 &a, const fe2 &b) {
        fe2 r;
        // (c0 and c1 as a dual-issue pair, Field::lmul2_x2, was measured: no gain -- a two-term dot product at 2 waves/SIMD already
        // runs at 91 % of its issue bound, profiles/r03_dual_issue.txt -- for 10 more VGPRs)
        r.c0 = Fq::lmul2(a.c0, b.c0, a.c1, Fq::lneg_op(b.c1));
        r.c1 = Fq::lmul2(a.c0, b.c1, a.c1, b.c0);
        return r;
    }
    static constexpr bool PAIRS = false;                              // the G2 formulas keep their order (see lmul)
    static ZK_HD void lmul_pair(const fe2 &a, const fe2 &b, const fe2 &c, const fe2 &d, fe2 &r0, fe2 &r1) { r0 = lmul(a, b); r1 = lmul(c, d); }
    static ZK_HD void lsqr_pair(const fe2 &a, const fe2 &b, fe2 &r0, fe2 &r1) { r0 = lsqr(a); r1 = lsqr(b); }
    // a*b + c*d in Fq2, one reduction per component
    static ZK_HD fe2 lmul2(const fe2 &a, const fe2 &b, const fe2 &c, const fe2 &d) {
        fe2 r;
        r.c0 = Fq::lmul4(a.c0, b.c0, a.c1, Fq::lneg_op(b.c1), c.c0, d.c0, c.c1, Fq::lneg_op(d.c1));
        r.c1 = Fq::lmul4(a.c0, b.c1, a.c1, b.c0, c.c0, d.c1, c.c1, d.c0);
        return r;
    }
    static ZK_HD fe2 lsqr(const fe2 &a) {
        fe p = Fq::lmul(a.c0, a.c1);
        fe2 r; r.c0 = Fq::lmul(Fq::ladd(a.c0, a.c1), Fq::lsub(a.c0, a.c1)); r.c1 = Fq::ladd(p, p); return r;
    }
    static ZK_HD fe2 canon(const fe2 &a) { fe2 r; r.c0 = Fq::canon(a.c0); r.c1 = Fq::canon(a.c1); return r; }
    static ZK_HD fe2 qsel(uint32_t ql, const fe2 &a, const fe2 &b, const fe2 &c, const fe2 &d) {
        fe2 r; r.c0 = Fq::qsel(ql, a.c0, b.c0, c.c0, d.c0); r.c1 = Fq::qsel(ql, a.c1, b.c1, c.c1, d.c1); return r;
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
    template <int S> static __device__ __forceinline__ fe2 qbcast(const fe2 &v) { fe2 r; r.c0 = Fq::qbcast<S>(v.c0); r.c1 = Fq::qbcast<S>(v.c1); return r; }
#endif
    static ZK_HD fe2 inv(const fe2 &a) {
        fe n = Fq::inv(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
        fe2 r; r.c0 = Fq::mul(a.c0, n); r.c1 = Fq::neg(Fq::mul(a.c1, n)); return r;
    }
};

// ------------------------------------------------------------------------------------------------
// short Weierstrass y^2 = x^3 + b, a = 0, over F (Fq for G1, Fq2 for G2); formulas never use b.
template <class F>
struct Curve {
    typedef typename F::elem E;
    // VGPR budget: 256 (G2, 2 waves/SIMD), 128 (G1, 4 waves/SIMD).  The machine-filling G1 accumulations use the mixed addition with
    // dual-issue product pairs (madd_pairs), which needs 156 VGPRs: 3 waves/SIMD (any pair at 4 waves/SIMD spills).  Same-box A/B at
    // 2^20, accumulation kernels alone: 1.31 -> 1.245 ms per G1 query, bench +1.4 % (profiles/r03_dual_issue.txt); latency-sized
    // multi-exponentiations keep the plain form at 4 waves/SIMD (fewer resident threads mean longer chunks: synchronous 2^17 proof
    // 2.38 -> 2.59 ms with the pairs form) -- MsmShape::acc_pairs.
    static constexpr int WAVES_PER_SIMD = sizeof(E) > 32 ? 2 : 4;
    static constexpr int WAVES_PER_SIMD_PAIRS = sizeof(E) > 32 ? 2 : 3;
    struct alignas(16) Affine { E x, y; };
    struct alignas(16) XYZZ { E X, Y, ZZ, ZZZ; };

    static ZK_HD bool is_inf(const Affine &p) { return F::lis_zero(p.x) && F::lis_zero(p.y); }
    static ZK_HD bool is_inf(const XYZZ &p) { return F::lis_zero(p.ZZ); }
    static ZK_HD XYZZ infinity() { XYZZ r; r.X = F::zero(); r.Y = F::zero(); r.ZZ = F::zero(); r.ZZZ = F::zero(); return r; }
    static ZK_HD Affine aff_infinity() { Affine r; r.x = F::zero(); r.y = F::zero(); return r; }
    static ZK_HD XYZZ from_affine(const Affine &p) {
        if (is_inf(p)) return infinity();
        XYZZ r; r.X = p.x; r.Y = p.y; r.ZZ = F::one(); r.ZZZ = F::one(); return r;
    }
    static ZK_HD Affine neg(const Affine &p) { Affine r; r.x = p.x; r.y = F::lneg(p.y); return r; }
    static ZK_HD XYZZ neg(const XYZZ &p) { XYZZ r = p; r.Y = F::lneg(p.Y); return r; }

    // 2 * affine (mdbl-2008-s-1)
    static ZK_HD XYZZ dbl_affine(const Affine &p) {
        if (is_inf(p)) return infinity();
        E U = F::ldbl(p.y), V = F::lsqr(U), W = F::lmul(U, V), S = F::lmul(p.x, V);
        E xx = F::lsqr(p.x), M = F::ladd(F::ldbl(xx), xx);
        XYZZ r;
        r.X = F::lsub(F::lsqr(M), F::ldbl(S));
        r.Y = F::lmul2(M, F::lsub(S, r.X), W, F::lneg_yop(p.y));           // M (S - X3) - W y: one reduction
        r.ZZ = V; r.ZZZ = W;
        return r;
    }
    // dbl-2008-s-1
    static ZK_HD XYZZ dbl(const XYZZ &p) {
        if (is_inf(p)) return p;
        E U = F::ldbl(p.Y), V = F::lsqr(U), W = F::lmul(U, V), S = F::lmul(p.X, V);
        E xx = F::lsqr(p.X), M = F::ladd(F::ldbl(xx), xx);
        XYZZ r;
        r.X = F::lsub(F::lsqr(M), F::ldbl(S));
        r.Y = F::lmul2(M, F::lsub(S, r.X), W, F::lneg_yop(p.Y));
        r.ZZ = F::lmul(V, p.ZZ); r.ZZZ = F::lmul(W, p.ZZZ);
        return r;
    }
    // p + q, q affine (madd-2008-s); all exceptional cases handled -- bit-exactness needs them
    static ZK_HD XYZZ madd(const XYZZ &p, const Affine &q) {
        if (is_inf(q)) return p;
        if (is_inf(p)) { XYZZ r; r.X = q.x; r.Y = q.y; r.ZZ = F::one(); r.ZZZ = F::one(); return r; }
        E U2 = F::lmul(q.x, p.ZZ), S2 = F::lmul(q.y, p.ZZZ);
        E Pd = F::lsub(U2, p.X), R = F::lsub(S2, p.Y);
        if (F::lis_zero(Pd)) {
            if (F::lis_zero(R)) return dbl_affine(q);
            return infinity();
        }
        XYZZ r;                                                        // operands die as early as possible (register pressure)
        const E PP = F::lsqr(Pd);
        r.ZZ = F::lmul(p.ZZ, PP);
        const E Q = F::lmul(p.X, PP), PPP = F::lmul(Pd, PP);
        r.ZZZ = F::lmul(p.ZZZ, PPP);
        r.X = F::lsub(F::lsub(F::lsqr(R), PPP), F::ldbl(Q));
        r.Y = F::lmul2(R, F::lsub(Q, r.X), F::lneg_yop(p.Y), PPP);        // R (Q - X3) - Y1 PPP: one reduction
        return r;
    }
    // the same sum with the independent products of the formula issued in pairs (Field::lmul_pair: dual-issue statements over Fq;
    // over Fq2 -- and on the host -- it is madd)
    static ZK_HD XYZZ madd_pairs(const XYZZ &p, const Affine &q) {
        if constexpr (!F::PAIRS) return madd(p, q);
        else {
        if (is_inf(q)) return p;
        if (is_inf(p)) { XYZZ r; r.X = q.x; r.Y = q.y; r.ZZ = F::one(); r.ZZZ = F::one(); return r; }
        E U2, S2;
        F::lmul_pair(q.x, p.ZZ, q.y, p.ZZZ, U2, S2);
        E Pd = F::lsub(U2, p.X), R = F::lsub(S2, p.Y);
        if (F::lis_zero(Pd)) {
            if (F::lis_zero(R)) return dbl_affine(q);
            return infinity();
        }
        XYZZ r;
        E PP, RR, Q, PPP;
        F::lsqr_pair(Pd, R, PP, RR);
        F::lmul_pair(p.X, PP, Pd, PP, Q, PPP);
        F::lmul_pair(p.ZZ, PP, p.ZZZ, PPP, r.ZZ, r.ZZZ);
        r.X = F::lsub(F::lsub(RR, PPP), F::ldbl(Q));
        r.Y = F::lmul2(R, F::lsub(Q, r.X), F::lneg_yop(p.Y), PPP);
        return r;
        }
    }
    // p + q (add-2008-s)
    static ZK_HD XYZZ add(const XYZZ &p, const XYZZ &q) {
        if (is_inf(p)) return q;
        if (is_inf(q)) return p;
        E U1 = F::lmul(p.X, q.ZZ), U2 = F::lmul(q.X, p.ZZ);
        E S1 = F::lmul(p.Y, q.ZZZ), S2 = F::lmul(q.Y, p.ZZZ);
        E Pd = F::lsub(U2, U1), R = F::lsub(S2, S1);
        if (F::lis_zero(Pd)) {
            if (F::lis_zero(R)) return dbl(p);
            return infinity();
        }
        E PP = F::lsqr(Pd), PPP = F::lmul(Pd, PP), Q = F::lmul(U1, PP);
        XYZZ r;
        r.X = F::lsub(F::lsub(F::lsqr(R), PPP), F::ldbl(Q));
        r.Y = F::lmul2(R, F::lsub(Q, r.X), F::lneg_yop(S1), PPP);
        r.ZZ = F::lmul(F::lmul(p.ZZ, q.ZZ), PP); r.ZZZ = F::lmul(F::lmul(p.ZZZ, q.ZZZ), PPP);
        return r;
    }
    // ---- quad-cooperative forms.  Latency-bound kernels (the bucket reductions of small MSMs) run one logical thread
    // on FOUR adjacent lanes that hold the same operands; the independent field products of a formula are dealt to
    // the lanes round by round (lane ql takes product ql) and handed back with a DPP quad broadcast, so an addition
    // costs 4 product latencies instead of 14, a doubling 3 instead of 9, a mixed addition 4 instead of 10.  Every
    // lane returns the full result; control flow is uniform across the quad (same data).  Host code and the CPU
    // emulation have no lanes: there the _q forms are the plain ones.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM)
#define ZK_QB(S, v) F::template qbcast<S>(v)
    static __device__ __forceinline__ XYZZ dbl_q(const XYZZ &p, uint32_t ql) {
        if (is_inf(p)) return p;
        const E U = F::ldbl(p.Y);
        E a = F::qsel(ql, U, p.X, U, U);
        E t = F::lmul(a, a);                                                     // V = U^2 | xx = X^2
        const E V = ZK_QB(0, t), xx = ZK_QB(1, t), M = F::ladd(F::ldbl(xx), xx);
        t = F::lmul(F::qsel(ql, U, p.X, M, V), F::qsel(ql, V, V, M, p.ZZ));      // W = U V | S = X V | M^2 | ZZ3 = V ZZ
        const E W = ZK_QB(0, t), S = ZK_QB(1, t), MM = ZK_QB(2, t);
        XYZZ r; r.ZZ = ZK_QB(3, t);
        r.X = F::lsub(MM, F::ldbl(S));
        t = F::lmul(F::qsel(ql, M, W, W, W), F::qsel(ql, F::lsub(S, r.X), p.Y, p.ZZZ, p.ZZZ));   // M (S - X3) | W Y | ZZZ3 = W ZZZ
        r.Y = F::lsub(ZK_QB(0, t), ZK_QB(1, t)); r.ZZZ = ZK_QB(2, t);
        return r;
    }
    static __device__ __forceinline__ XYZZ add_q(const XYZZ &p, const XYZZ &q, uint32_t ql) {
        if (is_inf(p)) return q;
        if (is_inf(q)) return p;
        E t = F::lmul(F::qsel(ql, p.X, q.X, p.Y, q.Y), F::qsel(ql, q.ZZ, p.ZZ, q.ZZZ, p.ZZZ));   // U1 | U2 | S1 | S2
        const E U1 = ZK_QB(0, t), U2 = ZK_QB(1, t), S1 = ZK_QB(2, t), S2 = ZK_QB(3, t);
        const E Pd = F::lsub(U2, U1), R = F::lsub(S2, S1);
        if (F::lis_zero(Pd)) {
            if (F::lis_zero(R)) return dbl_q(p, ql);
            return infinity();
        }
        t = F::lmul(F::qsel(ql, Pd, R, p.ZZ, p.ZZZ), F::qsel(ql, Pd, R, q.ZZ, q.ZZZ));          // PP | R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
        const E PP = ZK_QB(0, t), RR = ZK_QB(1, t), ZZ12 = ZK_QB(2, t), ZZZ12 = ZK_QB(3, t);
        t = F::lmul(F::qsel(ql, Pd, U1, ZZ12, ZZ12), PP);                                        // PPP | Q | ZZ3
        const E PPP = ZK_QB(0, t), Q = ZK_QB(1, t);
        XYZZ r; r.ZZ = ZK_QB(2, t);
        r.X = F::lsub(F::lsub(RR, PPP), F::ldbl(Q));
        t = F::lmul(F::qsel(ql, R, S1, ZZZ12, ZZZ12), F::qsel(ql, F::lsub(Q, r.X), PPP, PPP, PPP));   // R (Q - X3) | S1 PPP | ZZZ3
        r.Y = F::lsub(ZK_QB(0, t), ZK_QB(1, t)); r.ZZZ = ZK_QB(2, t);
        return r;
    }
    static __device__ __forceinline__ XYZZ madd_q(const XYZZ &p, const Affine &q, uint32_t ql) {
        if (is_inf(q)) return p;
        if (is_inf(p)) { XYZZ r; r.X = q.x; r.Y = q.y; r.ZZ = F::one(); r.ZZZ = F::one(); return r; }
        E t = F::lmul(F::qsel(ql, q.x, q.y, q.x, q.y), F::qsel(ql, p.ZZ, p.ZZZ, p.ZZ, p.ZZZ));   // U2 | S2
        const E Pd = F::lsub(ZK_QB(0, t), p.X), R = F::lsub(ZK_QB(1, t), p.Y);
        if (F::lis_zero(Pd)) {
            if (F::lis_zero(R)) return dbl_affine(q);
            return infinity();
        }
        E a = F::qsel(ql, Pd, R, Pd, R);
        t = F::lmul(a, a);                                                                       // PP | R^2
        const E PP = ZK_QB(0, t), RR = ZK_QB(1, t);
        t = F::lmul(F::qsel(ql, Pd, p.X, p.ZZ, p.ZZ), PP);                                       // PPP | Q | ZZ3
        const E PPP = ZK_QB(0, t), Q = ZK_QB(1, t);
        XYZZ r; r.ZZ = ZK_QB(2, t);
        r.X = F::lsub(F::lsub(RR, PPP), F::ldbl(Q));
        t = F::lmul(F::qsel(ql, R, p.Y, p.ZZZ, p.ZZZ), F::qsel(ql, F::lsub(Q, r.X), PPP, PPP, PPP));   // R (Q - X3) | Y1 PPP | ZZZ3
        r.Y = F::lsub(ZK_QB(0, t), ZK_QB(1, t)); r.ZZZ = ZK_QB(2, t);
        return r;
    }
#undef ZK_QB
#else
    static ZK_HD XYZZ dbl_q(const XYZZ &p, uint32_t) { return dbl(p); }
    static ZK_HD XYZZ add_q(const XYZZ &p, const XYZZ &q, uint32_t) { return add(p, q); }
    static ZK_HD XYZZ madd_q(const XYZZ &p, const Affine &q, uint32_t) { return madd(p, q); }
#endif
    // Q lanes per logical thread: the plain forms for Q = 1, the quad forms for Q = 4
    template <int Q> static ZK_HD XYZZ addQ(const XYZZ &p, const XYZZ &q, uint32_t ql) { if constexpr (Q == 4) return add_q(p, q, ql); else return add(p, q); }
    template <int Q> static ZK_HD XYZZ dblQ(const XYZZ &p, uint32_t ql) { if constexpr (Q == 4) return dbl_q(p, ql); else return dbl(p); }
    template <int Q> static ZK_HD XYZZ maddQ(const XYZZ &p, const Affine &q, uint32_t ql) { if constexpr (Q == 4) return madd_q(p, q, ql); else return madd(p, q); }
    // Q = 1: plain, 4: quad-cooperative, 2: one lane, product pairs (the machine-filling G1 accumulations)
    template <int Q> static ZK_HD XYZZ maddV(const XYZZ &p, const Affine &q, uint32_t ql) { if constexpr (Q == 2) return madd_pairs(p, q); else return maddQ<Q>(p, q, ql); }

    // The formulas above run in the loose domain [0, 2p) of the field (no-op on the host).  to_affine normalises
    // its input and computes strictly, so affine outputs (tables, keys) are canonical; canon() normalises an XYZZ
    // point that leaves the device as it is (MSM results).
    static ZK_HD XYZZ canon(const XYZZ &p) { XYZZ r; r.X = F::canon(p.X); r.Y = F::canon(p.Y); r.ZZ = F::canon(p.ZZ); r.ZZZ = F::canon(p.ZZZ); return r; }
    static ZK_HD Affine to_affine(const XYZZ &p) {      // one inversion: 1/(ZZ*ZZZ)
        if (is_inf(p)) return aff_infinity();
        const XYZZ c = canon(p);                        // strict arithmetic from here on
        E i = F::inv(F::mul(c.ZZ, c.ZZZ));
        E izz = F::mul(i, c.ZZZ), izzz = F::mul(i, c.ZZ);
        Affine r; r.x = F::mul(c.X, izz); r.y = F::mul(c.Y, izzz); return r;
    }
};

typedef Curve<Fq> G1;
typedef Curve<Fq2> G2;

}  // namespace zk
