// ntt.hpp -- Fr radix-2 NTT over the libfqfft evaluation domain {omega_m^j}, and the
// R1CS witness -> QAP "H polynomial" pipeline built on it.
//
// Replaces (reference call sites; the bodies live in the ABSENT libfqfft / libsnark forks):
//   get_domain -> basic_radix2_domain / recursive_domain      src/stubs.cpp:61-75
//   r1cs_to_qap_witness_map(domain, cs, w, aA, aB, aH)        r1cs_gg_ppzksnark_zok.tcc:461-468
// Semantics follow SURVEY.md Appendix A.2/A.3: FFT, iFFT (x 1/m), cosetFFT (g = 5), icosetFFT,
// divide_by_Z_on_coset.
//
// Kernel shape (32 B elements; bound by the Fr products of the butterflies -- VALU issue -- not by HBM): a transform
// of size m = 2^logm is done in passes; each pass loads a tile of 2^TILE_LOG elements into LDS (struct-of-limbs layout: 8 planes of u32,
// so a wave's 64 lanes hit 64 consecutive banks), runs k butterfly stages there -- two per LDS round trip, four elements per thread in
// registers -- and writes the tile back.  Pass 0 does the bit-reversal permutation on load, later passes gather NCOL-wide contiguous
// column groups (>= 256 B per row) so that strided stages still move whole cache lines.
// Scaling vectors (1/m, coset powers) are fused into the first load / last store.
#pragma once
#include "bn254.hpp"
#include "common.hpp"

namespace zk {

#ifndef ZK_NTT_TILE_LOG
#define ZK_NTT_TILE_LOG 11
#endif
constexpr int NTT_TILE_LOG = ZK_NTT_TILE_LOG;    // 2048 elements = 64 KiB of LDS per workgroup
constexpr int NTT_TILE = 1 << NTT_TILE_LOG;
// threads per workgroup: every thread keeps FOUR elements in registers through two stages (k_ntt_pass), so 512 threads cover the 2048-element
// tile; two workgroups per CU (64 KiB each) give 4 waves per SIMD at 92 VGPRs.  tools/ntt_bench, same box, six transforms at 2^20 / 2^18 / 2^22:
// 0.786 / 0.275 / 3.56 ms against 0.813 / 0.280 / 3.68 ms for one stage per LDS round trip with 1024 threads (round 3 until then; 1024 threads
// with the paired stages: 0.932 -- half of them idle in a stage pair; 256: 0.843; a 4096-element tile with 1024 threads, one workgroup per CU: 0.852).
// With the first stage pair fed from memory and the last step stored to it: 0.792 -> 0.763 / 0.276 -> 0.261 / 3.48 -> 3.24 ms (profiles/r03_ntt_alone.txt)
#ifndef ZK_NTT_THREADS
#define ZK_NTT_THREADS 512
#endif
constexpr int NTT_THREADS = ZK_NTT_THREADS;
constexpr int NTT_MIN_LOGC = 2;                  // >= 4 columns = 128 contiguous bytes (one cache line) per tile row

static ZK_HD uint32_t bitrev32(uint32_t v, uint32_t bits) {
#if defined(__clang__)
    return bits ? __builtin_bitreverse32(v) >> (32 - bits) : 0;          // v_bfrev_b32 + shift (bits <= 28)
#else
    uint32_t r = 0;                                                      // (g++: the CPU emulation of tests/emul)
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
#endif
}

// One pass = stages s0+1 .. s0+k of the decimation-in-time transform.
//   tile element e -> column c = e & (ncol-1), row q = e >> logc
//   global index    = (h << (s0+k)) | (q << s0) | (lb*ncol + c)        (h, lb from blockIdx)
//   src of pass 0   = bitrev(global index)      (ncol = 1, s0 = 0)
// pre[]  (nullable) multiplies input element with natural index i by pre[i]  (coset shift g^i)
// post[] (nullable) multiplies output element i by post[i]                   (1/m, g^-i/m, ...)
// blockIdx.y selects one of several equal-size vectors laid out `batch_stride` elements apart (the A, B and C
// polynomials of the witness map are transformed by one launch per pass)
// Fused steps of the witness map (NttFuse; all optional, wave-uniform branches):
//   in2      the transform's input is the pointwise product in[i] * in2[i]  (A * B on the coset; first pass only)
//   post_alt vectors blockIdx.y >= alt_from are scaled by post_alt[] instead of post[]  (the C polynomial of a batched inverse)
//   sub      out[i] = v * post[i] - sub[i]  (last pass; needs post)
struct NttFuse { const fe *in2 = nullptr; const fe *post_alt = nullptr; uint32_t alt_from = 0xffffffffu; const fe *sub = nullptr; };
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass(const fe *__restrict__ in, fe *__restrict__ out, const fe *__restrict__ tw,
           uint32_t logm, uint32_t s0, uint32_t k, uint32_t logc, int bitrev_load,
           const fe *__restrict__ pre, const fe *__restrict__ post, uint32_t batch_stride, NttFuse fuse, int last_pass) {
    __shared__ uint32_t sh[8][NTT_TILE];
    in += (size_t)blockIdx.y * batch_stride; out += (size_t)blockIdx.y * batch_stride;
    if (fuse.in2) fuse.in2 += (size_t)blockIdx.y * batch_stride;
    if (fuse.sub) fuse.sub += (size_t)blockIdx.y * batch_stride;
    if (fuse.post_alt && blockIdx.y >= fuse.alt_from) post = fuse.post_alt;
    const uint32_t tile = 1u << (k + logc), ncol = 1u << logc;
    const uint32_t nlb = (1u << s0) >> logc;                  // column groups per high index
    const uint32_t lb = blockIdx.x & (nlb - 1), h = blockIdx.x / nlb;
    const uint32_t base = (h << (s0 + k)) | (lb << logc);

    // element of the tile -> index in the vector; what a pass reads (fetch) and writes (emit) there
    auto gidx = [&](uint32_t e) -> uint32_t { return base | ((e >> logc) << s0) | (e & (ncol - 1)); };
    auto fetch = [&](uint32_t idx) -> fe {
        const uint32_t src = bitrev_load ? bitrev32(idx, logm) : idx;
        fe v = in[src];
        if (fuse.in2) v = Fr::mul(v, fuse.in2[src]);
        if (pre) v = Fr::mul(v, pre[src]);
        return v;
    };
    auto emit = [&](uint32_t idx, fe v) {
        if (post) v = Fr::mul(v, post[idx]);                    // strict product: canonical whatever the (loose) input
        else if (last_pass) v = Fr::canon(v);
        if (fuse.sub) v = Fr::sub(v, fuse.sub[idx]);
        out[idx] = v;
    };
    // The FIRST stage pair takes its four elements straight from memory and the LAST step (a stage pair, or the single stage an odd count ends
    // with) stores straight to it: two LDS round trips and two barriers less per pass (pass 0 of 2^20: 5 instead of 7, the later pass 4 instead of
    // 6).  Later passes work in place; a workgroup has read its whole tile before its last step stores.  k < 2: through the tile as before.
    const bool load_fused = k >= 2;
    if (!load_fused) {
        for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
            const fe v = fetch(gidx(e));
#pragma unroll
            for (int l = 0; l < 8; l++) sh[l][e] = v.l[l];
        }
        __syncthreads();
    }
    // Stages in PAIRS, four elements per thread in registers (radix 4 with the radix-2 twiddle table): rows q00 < q01 < q10 < q11 differ in bits
    // s - 1 and s of the row index; stage s multiplies rows q01, q11 by T1 = w_S^j, stage s + 1 rows q10, q11 by T2a = w_{S+1}^j and
    // T2b = w_{S+1}^(j + 2^(S-1)) = tw[index of T2a + m / 4].  Half the LDS traffic, barriers and index arithmetic of one stage per round trip.
    uint32_t s = 1;
    for (; s + 1 <= k; s += 2) {
        const uint32_t half = 1u << (s - 1), S = s0 + s;
        const bool from_mem = s == 1, to_mem = s + 1 == k;           // (uniform)
        for (uint32_t qd = threadIdx.x; qd < (tile >> 2); qd += blockDim.x) {
            const uint32_t c = qd & (ncol - 1), r = qd >> logc;
            const uint32_t jm = r & (half - 1), grp = r >> (s - 1);
            const uint32_t e00 = (((grp << (s + 1)) | jm) << logc) | c, e01 = e00 + (half << logc), e10 = e00 + (half << (logc + 1)), e11 = e10 + (half << logc);
            const uint32_t j = (jm << s0) | ((lb << logc) | c);
            const size_t i2 = (size_t)j << (logm - S - 1);             // index of T2a; T1 = T2a^2 sits at 2 i2, T2b at i2 + m / 4
            fe x0, x1, x2, x3;
            if (from_mem) { x0 = fetch(gidx(e00)); x1 = fetch(gidx(e01)); x2 = fetch(gidx(e10)); x3 = fetch(gidx(e11)); }
            else {
#pragma unroll
                for (int l = 0; l < 8; l++) { x0.l[l] = sh[l][e00]; x1.l[l] = sh[l][e01]; x2.l[l] = sh[l][e10]; x3.l[l] = sh[l][e11]; }
            }
#ifdef ZK_NTT_NOPAIR
            if (S > 1) { const fe t1 = tw[i2 << 1]; x1 = Fr::lmul(x1, t1); x3 = Fr::lmul(x3, t1); }
#else
            if (S > 1) { const fe t1 = tw[i2 << 1]; Fr::lmul_pair(x1, t1, x3, t1, x1, x3); }      // stage 1: every twiddle is w^0 = 1 (uniform branch)
#endif
            fe a0 = Fr::ladd(x0, x1), a1 = Fr::lsub(x0, x1), a2 = Fr::ladd(x2, x3), a3 = Fr::lsub(x2, x3);
            if (S == 1) a3 = Fr::lmul(a3, tw[(size_t)1 << (logm - 2)]);                            // stages 1 and 2 of the transform: T2a = w^0 = 1 as well
            else {
#ifdef ZK_NTT_NOPAIR
                a2 = Fr::lmul(a2, tw[i2]); a3 = Fr::lmul(a3, tw[i2 + ((size_t)1 << (logm - 2))]);
#else
                Fr::lmul_pair(a2, tw[i2], a3, tw[i2 + ((size_t)1 << (logm - 2))], a2, a3);
#endif
            }
            x0 = Fr::ladd(a0, a2); x2 = Fr::lsub(a0, a2); x1 = Fr::ladd(a1, a3); x3 = Fr::lsub(a1, a3);
            if (to_mem) { emit(gidx(e00), x0); emit(gidx(e01), x1); emit(gidx(e10), x2); emit(gidx(e11), x3); }
            else {
#pragma unroll
                for (int l = 0; l < 8; l++) { sh[l][e00] = x0.l[l]; sh[l][e01] = x1.l[l]; sh[l][e10] = x2.l[l]; sh[l][e11] = x3.l[l]; }
            }
        }
        if (!to_mem) __syncthreads();
    }
    if (s <= k) {                                                    // an odd stage count ends with one radix-2 stage, stored straight to memory
        const uint32_t half = 1u << (s - 1), S = s0 + s;
        for (uint32_t bf = threadIdx.x; bf < (tile >> 1); bf += blockDim.x) {
            uint32_t c = bf & (ncol - 1), q = bf >> logc;
            uint32_t jm = q & (half - 1), grp = q >> (s - 1);
            uint32_t eu = (((grp << s) | jm) << logc) | c, ev = eu + (half << logc);
            uint32_t j = (jm << s0) | ((lb << logc) | c);      // index inside the half-group of stage S
            fe u, v;
#pragma unroll
            for (int l = 0; l < 8; l++) { u.l[l] = sh[l][eu]; v.l[l] = sh[l][ev]; }
            // butterflies in the loose domain [0, 2p) (bn254.hpp: no conditional subtraction behind the product, sums and differences
            // fold by 2p): 25 instructions less per butterfly; the last pass normalises what it stores
            if (S > 1) v = Fr::lmul(v, tw[(size_t)j << (logm - S)]);   // stage 1: every twiddle is w^0 = 1 (uniform branch)
            emit(gidx(eu), Fr::ladd(u, v)); emit(gidx(ev), Fr::lsub(u, v));
        }
    } else if (k == 0) {                                             // a size-1 tile: scaling only
        for (uint32_t e = threadIdx.x; e < tile; e += blockDim.x) {
            fe v;
#pragma unroll
            for (int l = 0; l < 8; l++) v.l[l] = sh[l][e];
            emit(gidx(e), v);
        }
    }
}

// tables: tw[i] = w^i (i < m/2); geo[i] = s * g^i (i < m).  One thread per 256-element run.
__global__ void k_fill_geometric(fe *out, uint32_t n, fe g, fe s) {
    uint32_t run = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i0 = run * 256;
    if (i0 >= n) return;
    fe t = Fr::mul(Fr::pow_u64(g, i0), s);
    uint32_t end = i0 + 256 < n ? i0 + 256 : n;
    for (uint32_t i = i0; i < end; i++) { out[i] = t; t = Fr::mul(t, g); }
}

struct NttTables {
    uint32_t logm = 0;
    fe *tw_fwd = nullptr, *tw_inv = nullptr;   // m/2 each
    fe *coset_fwd = nullptr;                   // g^i            (cosetFFT pre-scale)
    fe *inv_m = nullptr;                       // 1/m            (iFFT post-scale)
    fe *icoset = nullptr;                      // g^-i / m       (icosetFFT post-scale)
    fe *inv_then_coset = nullptr;              // g^i / m        (iFFT followed by cosetFFT, fused)
    fe *inv_m_zinv = nullptr;                  // zinv / m       (iFFT of C, pre-divided by Z on the coset)
    fe *icoset_zinv = nullptr;                 // g^-i zinv / m  (icosetFFT and the division by Z on the coset, fused)
    fe zinv;                                   // (g^m - 1)^-1
};

static inline fe fr_domain_root(uint32_t logm) {
    // omega_{2^28} = 5^((r-1)/2^28)  (libff alt_bn128 Fr root_of_unity), canonical limbs
    fe w; const uint32_t c[8] = {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu, 0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u};
    for (int i = 0; i < 8; i++) w.l[i] = c[i];
    w = Fr::to_mont(w);
    for (uint32_t i = logm; i < 28; i++) w = Fr::sqr(w);
    return w;
}

static inline int ntt_tables_free(NttTables &t) {
    if (t.tw_fwd) hipFree(t.tw_fwd);
    if (t.tw_inv) hipFree(t.tw_inv);
    if (t.coset_fwd) hipFree(t.coset_fwd);
    if (t.inv_m) hipFree(t.inv_m);
    if (t.icoset) hipFree(t.icoset);
    if (t.inv_then_coset) hipFree(t.inv_then_coset);
    if (t.inv_m_zinv) hipFree(t.inv_m_zinv);
    if (t.icoset_zinv) hipFree(t.icoset_zinv);
    t = NttTables();
    return ZK_OK;
}

static inline int ntt_tables_create(NttTables &t, uint32_t logm, hipStream_t st) {
    if (logm > 28) return ZK_ERR_ARG;
    t.logm = logm;
    const uint32_t m = 1u << logm, half = m > 1 ? m / 2 : 1;
    ZK_HIP(hipMalloc(&t.tw_fwd, sizeof(fe) * half));
    ZK_HIP(hipMalloc(&t.tw_inv, sizeof(fe) * half));
    ZK_HIP(hipMalloc(&t.coset_fwd, sizeof(fe) * m));
    ZK_HIP(hipMalloc(&t.inv_m, sizeof(fe) * m));
    ZK_HIP(hipMalloc(&t.icoset, sizeof(fe) * m));
    ZK_HIP(hipMalloc(&t.inv_then_coset, sizeof(fe) * m));
    ZK_HIP(hipMalloc(&t.inv_m_zinv, sizeof(fe) * m));
    ZK_HIP(hipMalloc(&t.icoset_zinv, sizeof(fe) * m));
    const fe w = fr_domain_root(logm), wi = Fr::inv(w), g = Fr::from_u64(5), gi = Fr::inv(g);
    const fe mi = Fr::inv(Fr::from_u64(m)), one = Fr::one();
    t.zinv = Fr::inv(Fr::sub(Fr::pow_u64(g, m), one));
    auto fill = [&](fe *dst, uint32_t n, const fe &base, const fe &scale) {
        uint32_t runs = zk_div_up(n, 256);
        ZK_LAUNCH(k_fill_geometric, zk_div_up(runs, 64), 64, st, dst, n, base, scale);
    };
    fill(t.tw_fwd, half, w, one);
    fill(t.tw_inv, half, wi, one);
    fill(t.coset_fwd, m, g, one);
    fill(t.inv_m, m, one, mi);
    fill(t.icoset, m, gi, mi);
    fill(t.inv_then_coset, m, g, mi);
    fill(t.inv_m_zinv, m, one, Fr::mul(mi, t.zinv));
    fill(t.icoset_zinv, m, gi, Fr::mul(mi, t.zinv));
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// out <- transform(in); in and out must be different buffers (pass 0 permutes).  `pre`/`post` as in k_ntt_pass.
// batch > 1: `batch` vectors, `stride` elements apart in both in and out, one launch per pass.
static inline int ntt_run(const NttTables &t, const fe *in, fe *out, bool inverse,
                          const fe *pre, const fe *post, hipStream_t st, uint32_t batch = 1, uint32_t stride = 0, NttFuse fuse = NttFuse()) {
    const uint32_t logm = t.logm, m = 1u << logm;
    const fe *tw = inverse ? t.tw_inv : t.tw_fwd;
    if (logm == 0) {   // size-1 transform is the identity (times scaling)
        uint32_t k0 = 0;
        ZK_LAUNCH_SYNC(k_ntt_pass, dim3(1, batch), NTT_THREADS, st, in, out, tw, logm, 0u, k0, 0u, 1, pre, post, stride, fuse, 1);
        return ZK_OK;
    }
    const uint32_t k0 = logm < (uint32_t)NTT_TILE_LOG ? logm : NTT_TILE_LOG;
    uint32_t rem = logm - k0;
    const uint32_t kmax = NTT_TILE_LOG - NTT_MIN_LOGC;
    uint32_t npass = (rem + kmax - 1) / kmax;
    NttFuse first = fuse, later = fuse;                        // in2 belongs to the first pass, post / post_alt / sub to the last
    later.in2 = nullptr;
    if (rem != 0) { first.post_alt = nullptr; first.sub = nullptr; }
    ZK_LAUNCH_SYNC(k_ntt_pass, dim3(m >> k0, batch), NTT_THREADS, st, in, out, tw, logm, 0u, k0, 0u, 1, pre,
                   rem == 0 ? post : (const fe *)nullptr, stride, first, rem == 0 ? 1 : 0);
    uint32_t s0 = k0;
    while (rem) {
        uint32_t k = (rem + npass - 1) / npass;
        uint32_t logc = NTT_TILE_LOG - k;
        if (logc > s0) logc = s0;
        rem -= k; npass--;
        NttFuse f = later;
        if (rem != 0) { f.post_alt = nullptr; f.sub = nullptr; }
        ZK_LAUNCH_SYNC(k_ntt_pass, dim3(m >> (k + logc), batch), NTT_THREADS, st, (const fe *)out, out, tw, logm, s0, k, logc, 0,
                       (const fe *)nullptr, rem == 0 ? post : (const fe *)nullptr, stride, f, rem == 0 ? 1 : 0);
        s0 += k;
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// -------------------------------------------------------------------------------------------------
// Row evaluation  a_j = <A_j, w>  over CSR (thread per short row; long rows go through chunks).
constexpr uint32_t SPMV_LONG_ROW = 64;        // rows with more terms are split into chunks
constexpr uint32_t SPMV_CHUNK = 4096;         // terms per chunk (one workgroup each)

static ZK_HD bool fr_is_one(const fe &a) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t |= a.l[i] ^ FrParams::one(i);
    return t == 0;
}
static ZK_D fe spmv_term(const fe &coef, const fe &x) { return fr_is_one(coef) ? x : Fr::mul(coef, x); }

// blockIdx.y = proof of a batch: its witness lies w_stride elements, its output out_stride elements further on
// The same launch writes the PADDING of the polynomial: rows [n_rows, m_rows) are zero, except -- the A polynomial, n_in_rows = nIn + 1 --
// the input-consistency rows aA[nC + i] = w[i] (Appendix A.3 step 1).  (A memset and a kernel of their own until round 3.)
__global__ void k_spmv_rows(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
                            const fe *__restrict__ coeff, const fe *__restrict__ w, fe *__restrict__ out, uint32_t n_rows,
                            uint32_t w_stride, uint32_t out_stride, uint32_t m_rows, uint32_t n_in_rows) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m_rows) return;
    w += (size_t)blockIdx.y * w_stride; out += (size_t)blockIdx.y * out_stride;
    if (j >= n_rows) { out[j] = j - n_rows < n_in_rows ? w[j - n_rows] : Fr::zero(); return; }
    uint32_t b = row_ptr[j], e = row_ptr[j + 1];
    if (e - b > SPMV_LONG_ROW) return;                     // written by k_spmv_long_finish
    fe acc = Fr::zero();
    for (uint32_t k = b; k < e; k++) acc = Fr::add(acc, spmv_term(coeff[k], w[col[k]]));
    out[j] = acc;
}

// one workgroup per chunk of a long row: strided partial sums, then an LDS tree
__global__ void __launch_bounds__(256)
k_spmv_long_chunks(const uint32_t *__restrict__ chunk_begin, const uint32_t *__restrict__ chunk_end,
                   const uint32_t *__restrict__ col, const fe *__restrict__ coeff, const fe *__restrict__ w,
                   fe *__restrict__ partial, uint32_t w_stride) {
    __shared__ uint32_t sh[8][256];
    w += (size_t)blockIdx.y * w_stride; partial += (size_t)blockIdx.y * gridDim.x;
    const uint32_t b = chunk_begin[blockIdx.x], e = chunk_end[blockIdx.x];
    fe acc = Fr::zero();
    for (uint32_t k = b + threadIdx.x; k < e; k += blockDim.x) acc = Fr::add(acc, spmv_term(coeff[k], w[col[k]]));
#pragma unroll
    for (int l = 0; l < 8; l++) sh[l][threadIdx.x] = acc.l[l];
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            fe o;
#pragma unroll
            for (int l = 0; l < 8; l++) o.l[l] = sh[l][threadIdx.x + s];
            acc = Fr::add(acc, o);
#pragma unroll
            for (int l = 0; l < 8; l++) sh[l][threadIdx.x] = acc.l[l];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// workgroup per long row: sum its chunk partials (chunks of a row are consecutive) -- strided partial sums, then an LDS tree.
// (One THREAD per row summed 256 partials through 256 dependent loads and additions: 53 us per launch at 2^20.)
__global__ void __launch_bounds__(256)
k_spmv_long_finish(const uint32_t *__restrict__ long_row, const uint32_t *__restrict__ long_first_chunk,
                   const fe *__restrict__ partial, fe *__restrict__ out, uint32_t n_long, uint32_t n_chunks, uint32_t out_stride) {
    __shared__ uint32_t sh[8][256];
    const uint32_t i = blockIdx.x;
    if (i >= n_long) return;
    partial += (size_t)blockIdx.y * n_chunks; out += (size_t)blockIdx.y * out_stride;
    fe acc = Fr::zero();
    for (uint32_t k = long_first_chunk[i] + threadIdx.x; k < long_first_chunk[i + 1]; k += blockDim.x) acc = Fr::add(acc, partial[k]);
#pragma unroll
    for (int l = 0; l < 8; l++) sh[l][threadIdx.x] = acc.l[l];
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            fe o;
#pragma unroll
            for (int l = 0; l < 8; l++) o.l[l] = sh[l][threadIdx.x + s];
            acc = Fr::add(acc, o);
#pragma unroll
            for (int l = 0; l < 8; l++) sh[l][threadIdx.x] = acc.l[l];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[long_row[i]] = acc;
}

}  // namespace zk
