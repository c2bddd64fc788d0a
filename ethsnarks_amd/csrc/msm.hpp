// msm.hpp -- multi-scalar multiplication sum_i s_i * P_i over alt_bn128 G1 / G2 on gfx950.
//
// Replaces (reference call sites; bodies in the ABSENT libff / libsnark forks):
//   kc_multi_exp_with_mixed_addition<G1|G2, Fr, multi_exp_method_BDLO12>   tcc:488-506  (A-, B-query)
//   libff::multi_exp<G1, Fr, BDLO12>                                       tcc:510-518  (H-query)
//   libff::multi_exp_with_mixed_addition<G1, Fr, BDLO12>                   tcc:522-530  (L-query)
// BDLO12 is the bucket (Pippenger) method; the group element it returns is unique, so any schedule
// of the additions gives the same proof bytes.
//
// GPU schedule (all integer VALU work, random 64/128-B base gathers from HBM):
//   1 k_msm_digits      scalar -> canonical -> signed c-bit digits; per-(window,bucket) histogram
//   2 k_msm_scan_*      exclusive scans: entries per bucket, and fixed-length *segments* per bucket
//   3 k_msm_scatter     counting-sort scatter of (base index, sign) by bucket
//   4 k_msm_accumulate  one thread per segment (<= SEG entries): XYZZ mixed additions
//   5 k_msm_bucket_finalize / k_msm_heavy   segments -> bucket sums (heavy buckets by a workgroup)
//   6 k_msm_group_reduce   running-sum over K buckets + small-scalar offset -> partial window sums
//   7 k_msm_sum_groups  (x passes) partials -> one XYZZ sum per window
// The c-bit window shifts (Horner over <= 128 window sums) and the affine normalisation are O(W)
// sequential group operations and run on the host from the same bn254.hpp.
// Zero scalars produce no entries; scalar 1 (and any other repeated value) lands in one bucket whose
// entries are cut into segments, so 0/1-heavy witnesses (the *_with_mixed_addition fast paths of the
// reference) stay load-balanced without special cases.
#pragma once
#include "bn254.hpp"
#include "common.hpp"

namespace zk {

constexpr uint32_t MSM_SEG = 32;          // entries per accumulation thread
constexpr uint32_t MSM_HEAVY = 32;        // buckets with more segments than this are reduced by a workgroup
constexpr uint32_t MSM_GROUP = 16;        // buckets per running-sum thread
constexpr uint32_t MSM_SUMW = 16;         // fan-in of the partial-sum passes
constexpr uint32_t MSM_KEY_NONE = 0xffffffffu;
#ifdef ZK_EMUL
constexpr uint32_t MSM_HEAVY_GRID = 2;    // the emulator spawns a real thread per GPU thread
#else
constexpr uint32_t MSM_HEAVY_GRID = 256;
#endif

// -------------------------------------------------------------------------------------------------
struct MsmShape {
    uint32_t n = 0, c = 0, W = 0, nb = 0, nbk = 0;
    static uint32_t pick_c(uint32_t n) {
        // accumulate ~ n*W mixed adds, reduce ~ W*2^(c-1) full adds on few threads: keep buckets well below n/8
        uint32_t c = 2;
        while (c < 16 && (1ull << (c + 4)) <= (uint64_t)n) c++;
        return c;
    }
    void set(uint32_t n_, uint32_t c_) {
        n = n_; c = c_ ? c_ : pick_c(n_);
        if (c < 2) c = 2;
        if (c > 20) c = 20;
        W = 254 / c + 1; nb = 1u << (c - 1); nbk = W * nb;
    }
    uint64_t max_entries() const { return (uint64_t)n * W; }
    uint64_t max_segments() const { return max_entries() / MSM_SEG + nbk + 1; }
};

template <class C>
struct MsmWork {
    MsmShape sh;
    uint32_t *keys = nullptr, *hist = nullptr, *off = nullptr, *segoff = nullptr, *cursor = nullptr, *sorted = nullptr;
    uint32_t *heavy_list = nullptr, *heavy_count = nullptr, *tile_a = nullptr, *tile_b = nullptr;
    typename C::XYZZ *segsum = nullptr, *bucket = nullptr, *partial_a = nullptr, *partial_b = nullptr;
    typename C::XYZZ *host_windows = nullptr;   // pinned, W entries
    hipEvent_t ev_acc0 = nullptr, ev_acc1 = nullptr;   // bracket k_msm_accumulate (the dominant kernel) on its stream
    float accumulate_ms() const { float ms = 0; if (ev_acc0 && ev_acc1) hipEventElapsedTime(&ms, ev_acc0, ev_acc1); return ms; }

    int alloc(uint32_t n, uint32_t c);
    void release();
    // enqueue the whole MSM on `st`; window sums land in host_windows after the stream drains
    int enqueue(const typename C::Affine *bases, const fe *scalars, const uint32_t *gather, uint32_t n,
                int canonical, hipStream_t st);

    // host: Horner over the window sums (after the stream has been synchronised)
    typename C::XYZZ finish() const {
        typename C::XYZZ acc = C::infinity();
        for (int w = (int)sh.W - 1; w >= 0; w--) {
            for (uint32_t k = 0; k < sh.c; k++) acc = C::dbl(acc);
            acc = C::add(acc, host_windows[w]);
        }
        return acc;
    }
};

}  // namespace zk
