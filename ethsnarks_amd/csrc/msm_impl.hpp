// msm_impl.hpp -- kernels and launch code of the multi-scalar multiplication (see msm.hpp for the
// schedule and the reference call sites).  Included only by msm_g1.cpp / msm_g2.cpp, which instantiate
// MsmWork<G1> / MsmWork<G2> so that the two curve instantiations compile in parallel.  Kernels that do
// not depend on the curve are still templated on it so that each translation unit owns its symbols.
#pragma once
#include "msm.hpp"
#include <string.h>

namespace zk {

// T[w][k] = 2^(c*w) * P_k, affine; thread per base (one-off, at context creation)
template <class C>
__global__ void __launch_bounds__(64, C::WAVES_PER_SIMD)
k_msm_precompute(const typename C::Affine *__restrict__ bases, uint32_t n, uint32_t c, uint32_t W,
                 typename C::Affine *__restrict__ table) {        // c: doublings from one row to the next (window bits x planes), W: rows
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const typename C::Affine p = bases[k];
    table[k] = p;
    typename C::XYZZ cur = C::from_affine(p);
    for (uint32_t w = 1; w < W; w++) {
        for (uint32_t j = 0; j < c; j++) cur = C::dbl(cur);
        table[(size_t)w * n + k] = C::to_affine(cur);
    }
}

// ---- bucket sort of the n*W digit entries, without global atomics ---------------------------------
// Pass 1 partitions the entries into CB <= 256 *coarse* bins (high bits of the bucket id): every workgroup
// owns a contiguous run of scalars, counts its entries per coarse bin in LDS (k_sort_count), a column scan
// turns the [bin][workgroup] count matrix into write positions (k_sort_colscan / k_sort_binscan), and the
// workgroup re-derives its digits and writes (bucket, payload) pairs at LDS-ranked positions
// (k_sort_partition): runs of ~64 consecutive pairs per bin per workgroup.  Pass 2 gives every coarse bin
// to one workgroup (k_sort_fine): LDS histogram of its <= 2^15/CB fine buckets, local scan, and an in-L2
// scatter of the payloads.  It also emits the bucket offsets off[0..nb].  payload = table index | sign.
// signed c-bit digit of window w (carry in/out); returns magnitude (0 = no entry) and sign
static ZK_D uint32_t msm_digit(const fe &s, uint32_t w, uint32_t c, uint32_t &carry, uint32_t &neg) {
    const uint32_t nb = 1u << (c - 1), full = 1u << c;
    uint32_t bit = w * c, limb = bit >> 5, off = bit & 31;
    uint32_t lo = 0, hi = 0;                                    // static selects: keeps the scalar in VGPRs
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) { if (k == limb) lo = s.l[k]; if (k == limb + 1) hi = s.l[k]; }
    uint64_t v = lo | ((uint64_t)hi << 32);
    uint32_t d = ((uint32_t)(v >> off) & (full - 1)) + carry;
    neg = 0; carry = 0;
    if (d > nb) { d = full - d; neg = 1; carry = 1; }         // digit in [-2^(c-1)+1, 2^(c-1)]
    return d;
}
// scalar il of proof p: Montgomery unless canonical
static ZK_D fe msm_load_scalar(const fe *__restrict__ scalars, const uint32_t *__restrict__ gather, uint32_t p, uint32_t stride, uint32_t il, int canonical) {
    fe s = scalars[(size_t)p * stride + (gather ? gather[il] : il)];
    return canonical ? s : Fr::from_mont(s);
}

// counts[bin * groups + group] = entries of this workgroup's scalars that fall into coarse bin `bin`
template <class C>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_count(const fe *__restrict__ scalars, const uint32_t *__restrict__ gather, uint32_t n, uint32_t batch, uint32_t stride, int canonical,
             uint32_t c, uint32_t W, uint32_t plog, SortShape ss, uint32_t *__restrict__ counts) {
    __shared__ uint32_t cnt[SORT_MAX_CB];
    for (uint32_t k = threadIdx.x; k < ss.cb; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    const uint32_t total = n * batch, nbp = 1u << (c - 1);     // scalars of all proofs; buckets per proof
    const uint32_t i0 = blockIdx.x * ss.per_group, i1 = (i0 + ss.per_group < total) ? i0 + ss.per_group : total;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const uint32_t p = i / n, il = i - p * n;
        const fe s = msm_load_scalar(scalars, gather, p, stride, il, canonical);
        uint32_t carry = 0, neg;
        for (uint32_t w = 0; w < W; w++) {
            uint32_t d = msm_digit(s, w, c, carry, neg);
            if (d) atomicAdd(&cnt[((((p << plog) | (w & ((1u << plog) - 1u))) * nbp) + d - 1) >> ss.fine_bits], 1u);      // bucket set = (proof, plane of the window)
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < ss.cb; k += blockDim.x) counts[(size_t)k * ss.groups + blockIdx.x] = cnt[k];
}

// one workgroup per coarse bin: exclusive scan of its row counts[bin][0..groups) in place; bin total out
template <class C>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_colscan(uint32_t *__restrict__ counts, uint32_t groups, uint32_t *__restrict__ bin_total) {
    __shared__ uint32_t sh[SORT_THREADS];
    uint32_t *row = counts + (size_t)blockIdx.x * groups;
    const uint32_t t = threadIdx.x, per = (groups + SORT_THREADS - 1) / SORT_THREADS;
    const uint32_t g0 = t * per, g1 = (g0 + per < groups) ? g0 + per : groups;
    uint32_t sum = 0;
    for (uint32_t g = g0; g < g1; g++) sum += row[g];
    sh[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
        uint32_t v = (t >= d) ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    uint32_t e = sh[t] - sum;
    for (uint32_t g = g0; g < g1; g++) { uint32_t v = row[g]; row[g] = e; e += v; }
    if (t == SORT_THREADS - 1) bin_total[blockIdx.x] = sh[t];
}

// single workgroup: bin_base[k] = sum of bin_total[0..k), bin_base[cb] = total entries
template <class C>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_binscan(const uint32_t *__restrict__ bin_total, uint32_t cb, uint32_t *__restrict__ bin_base) {
    __shared__ uint32_t sh[SORT_THREADS];
    const uint32_t t = threadIdx.x, v0 = t < cb ? bin_total[t] : 0;
    sh[t] = v0;
    __syncthreads();
    for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
        uint32_t v = (t >= d) ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    if (t < cb) bin_base[t] = sh[t] - v0;
    if (t == SORT_THREADS - 1) bin_base[cb] = sh[t];
}

// pass 1 scatter: (bucket, payload) pairs into the coarse-bin regions, LDS-ranked
template <class C>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_partition(const fe *__restrict__ scalars, const uint32_t *__restrict__ gather, uint32_t n, uint32_t batch, uint32_t stride, int canonical,
                 uint32_t c, uint32_t W, uint32_t plog, SortShape ss, const uint32_t *__restrict__ counts,
                 const uint32_t *__restrict__ bin_base, uint2 *__restrict__ pairs, uint32_t kbits) {
    __shared__ uint32_t pos[SORT_MAX_CB];
    for (uint32_t k = threadIdx.x; k < ss.cb; k += blockDim.x) pos[k] = bin_base[k] + counts[(size_t)k * ss.groups + blockIdx.x];
    __syncthreads();
    const uint32_t total = n * batch, nbp = 1u << (c - 1);
    const uint32_t i0 = blockIdx.x * ss.per_group, i1 = (i0 + ss.per_group < total) ? i0 + ss.per_group : total;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const uint32_t pr = i / n, il = i - pr * n;
        const fe s = msm_load_scalar(scalars, gather, pr, stride, il, canonical);
        uint32_t carry = 0, neg;
        for (uint32_t w = 0; w < W; w++) {
            uint32_t d = msm_digit(s, w, c, carry, neg);
            if (!d) continue;
            const uint32_t b = ((pr << plog) | (w & ((1u << plog) - 1u))) * nbp + d - 1;     // (proof, plane, bucket)
            uint32_t p = atomicAdd(&pos[b >> ss.fine_bits], 1u);
            // payload: table index of 2^(cw) P_il (own sort: row * n + il, row = w >> plog), or -- a sort that drives OTHER queries' tables --
            // (row << kbits) | il, which their accumulation kernels take apart with a shift and a mask instead of a division; sign in bit 31
            const uint32_t row = w >> plog;
            uint2 e; e.x = (kbits ? (row << kbits) | il : row * n + il) | (neg << 31); e.y = b;
            pairs[p] = e;
        }
    }
}

// the same scatter STAGED through LDS: a workgroup of per_group <= SORT_STAGE / W scalars ranks its entries by bin in LDS first (its
// per-bin counts are the differences of the scanned count matrix), then copies the runs out -- consecutive lanes write consecutive pairs of one
// bin, whole cache lines instead of one 8-byte store per lane and line (the direct scatter: 82 us per 2^20 sort, 41 us of them the stores)
template <class C>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_partition_staged(const fe *__restrict__ scalars, const uint32_t *__restrict__ gather, uint32_t n, uint32_t batch, uint32_t stride, int canonical,
                        uint32_t c, uint32_t W, uint32_t plog, SortShape ss, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ bin_total,
                        const uint32_t *__restrict__ bin_base, uint2 *__restrict__ pairs, uint32_t kbits) {
    __shared__ uint2 stage[SORT_STAGE];
    __shared__ uint32_t lbase[SORT_MAX_CB], lcur[SORT_MAX_CB], gpos[SORT_MAX_CB], scan[SORT_THREADS];
    const uint32_t t = threadIdx.x, g = blockIdx.x;
    // entries of this workgroup per bin (SORT_MAX_CB == SORT_THREADS: one bin per thread), their exclusive scan, and where bin t's run goes
    uint32_t cnt = 0;
    if (t < ss.cb) {
        const uint32_t cur = counts[(size_t)t * ss.groups + g], nxt = g + 1 < ss.groups ? counts[(size_t)t * ss.groups + g + 1] : bin_total[t];
        cnt = nxt - cur; gpos[t] = bin_base[t] + cur;
    }
    scan[t] = cnt;
    __syncthreads();
    for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
        const uint32_t v = t >= d ? scan[t - d] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    lbase[t] = lcur[t] = scan[t] - cnt;
    const uint32_t local_total = scan[SORT_THREADS - 1];
    __syncthreads();
    const uint32_t total = n * batch, nbp = 1u << (c - 1);
    const uint32_t i0 = g * ss.per_group, i1 = (i0 + ss.per_group < total) ? i0 + ss.per_group : total;
    for (uint32_t i = i0 + t; i < i1; i += blockDim.x) {
        const uint32_t pr = i / n, il = i - pr * n;
        const fe sc = msm_load_scalar(scalars, gather, pr, stride, il, canonical);
        uint32_t carry = 0, neg;
        for (uint32_t w = 0; w < W; w++) {
            uint32_t d = msm_digit(sc, w, c, carry, neg);
            if (!d) continue;
            const uint32_t b = ((pr << plog) | (w & ((1u << plog) - 1u))) * nbp + d - 1;     // (proof, plane, bucket)
            const uint32_t row = w >> plog;
            uint2 e; e.x = (kbits ? (row << kbits) | il : row * n + il) | (neg << 31); e.y = b;     // payload as in k_sort_partition
            stage[atomicAdd(&lcur[b >> ss.fine_bits], 1u)] = e;
        }
    }
    __syncthreads();
    for (uint32_t sl = t; sl < local_total; sl += blockDim.x) {
        const uint2 e = stage[sl];
        const uint32_t bin = e.y >> ss.fine_bits;
        pairs[gpos[bin] + (sl - lbase[bin])] = e;
    }
}

// pass 2: one workgroup per coarse bin sorts its region by fine bucket; emits off[] (nb + 1 entries) and sorted[].
// 1024 threads per workgroup: the pass is latency-bound (two sweeps over ~60 K pairs), so it wants every wave
// slot of the CU it runs on.
template <class C>
__global__ void __launch_bounds__(SORT_FINE_THREADS)
k_sort_fine(const uint2 *__restrict__ pairs, const uint32_t *__restrict__ bin_base, SortShape ss, uint32_t nb,
            uint32_t *__restrict__ off, uint32_t *__restrict__ sorted) {
    __shared__ uint32_t cnt[SORT_MAX_FB], cur[SORT_MAX_FB], sh[SORT_FINE_THREADS];
    const uint32_t bin = blockIdx.x, r0 = bin_base[bin], r1 = bin_base[bin + 1], t = threadIdx.x, T = blockDim.x;
    const uint32_t fmask = ss.fb - 1;
    for (uint32_t k = t; k < ss.fb; k += T) cnt[k] = 0;
    __syncthreads();
    // four independent loads per thread and round trip: the pass is a chain of dependent global-load latencies (one workgroup per
    // CU walks ~60 K pairs), not bandwidth
    for (uint32_t e = r0 + t; e < r1; e += 4 * T) {
        uint32_t y[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) y[u] = e + u * T < r1 ? pairs[e + u * T].y : 0xffffffffu;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) if (e + u * T < r1) atomicAdd(&cnt[y[u] & fmask], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0..fb): thread-serial runs + Hillis-Steele over the run sums
    const uint32_t per = (ss.fb + T - 1) / T, k0 = t * per, k1 = (k0 + per < ss.fb) ? k0 + per : ss.fb;
    uint32_t sum = 0;
    for (uint32_t k = k0; k < k1; k++) sum += cnt[k];
    sh[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < T; d <<= 1) {
        uint32_t v = (t >= d) ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    uint32_t ex = sh[t] - sum;
    for (uint32_t k = k0; k < k1; k++) {
        const uint32_t b = bin * ss.fb + k;
        if (b < nb) off[b] = r0 + ex;                       // (the last coarse bin may reach past the last bucket)
        cur[k] = r0 + ex;
        ex += cnt[k];
    }
    if (bin + 1 == gridDim.x && t == 0) off[nb] = r1;       // all entries
    __syncthreads();
    for (uint32_t e = r0 + t; e < r1; e += 4 * T) {
        uint2 pr[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) if (e + u * T < r1) pr[u] = pairs[e + u * T];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) if (e + u * T < r1) sorted[atomicAdd(&cur[pr[u].y & fmask], 1u)] = pr[u].x;
    }
}

// thread (Q = 1) or quad of lanes (Q = 4, see Curve::madd_q) per chunk of rule.len(off[nb]) consecutive sorted entries.
// Chunk c starts inside bucket b0 (binary search in off[]); its running sum is flushed as "piece" c + b whenever the
// entries move on to another bucket b, and at its end: piece numbers grow along the entry list, the pieces of bucket b
// are off[b] / len + b ... (off[b+1] - 1) / len + b.
// Q = 2: one lane per chunk like Q = 1, the mixed addition with dual-issue product pairs (Curve::madd_pairs) at the occupancy its
// registers allow (G1: 3 waves/SIMD instead of 4)
template <class C, int Q>
__global__ void __launch_bounds__(64, Q == 2 ? C::WAVES_PER_SIMD_PAIRS : C::WAVES_PER_SIMD)
k_msm_accumulate(const typename C::Affine *__restrict__ table, const uint32_t *__restrict__ sorted,
                 const uint32_t *__restrict__ off, uint32_t nb, ChunkRule rule,
                 uint32_t remap_src, uint32_t remap_offset, const uint32_t *__restrict__ remap_pos, uint32_t n_dst, uint32_t remap_kbits,
                 typename C::XYZZ *__restrict__ piece) {
    constexpr uint32_t LANES = Q == 4 ? 4 : 1;
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, c = gt / LANES, ql = gt % LANES;
    const uint32_t total = off[nb], seg = rule.len(total);
    if ((uint64_t)c * seg >= total) return;
    const uint32_t begin = c * seg, end = (total - begin < seg) ? total : begin + seg;
    uint32_t lo = 0, hi = nb;                                  // largest b with off[b] <= begin: the bucket of entry `begin`
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (off[mid] <= begin) lo = mid; else hi = mid; }
    uint32_t b = lo, bucket_end = off[b + 1];
    typename C::XYZZ acc = C::infinity();
    // software pipeline over the chunk: the sorted entry (and, where the registers allow it -- G1 --, the table point)
    // of iteration e + 1 is requested before the mixed addition of iteration e, so the two dependent gather latencies
    // hide behind ~3000 VALU instructions instead of stalling the wave between additions
    constexpr bool PREFETCH_POINT = sizeof(typename C::Affine) <= 64;
    auto entry_index = [&](uint32_t p) -> uint32_t {             // table index of an entry, or 0xffffffff when it is not this query's
        uint32_t idx = p & 0x7fffffffu;
        if (remap_src) {                                        // entry of another query's sort: (window, scalar) -> own table
            uint32_t w, i;
            if (remap_kbits) { w = idx >> remap_kbits; i = idx & ((1u << remap_kbits) - 1u); }      // (w << kbits) | i: the shared witness sort
            else { w = idx / remap_src; i = idx - w * remap_src; }
            const uint32_t k = remap_pos ? remap_pos[i] : i - remap_offset;     // unsigned wrap / 0xffffffff = absent
            if (k >= n_dst) return 0xffffffffu;
            idx = w * n_dst + k;
        }
        return idx;
    };
    uint32_t p_next = sorted[begin];
    typename C::Affine q_next = C::aff_infinity();
    if constexpr (PREFETCH_POINT) {
        const uint32_t idx = entry_index(p_next); if (idx != 0xffffffffu) q_next = table[idx];
    }
    for (uint32_t e = begin; e < end; e++) {
        if (e >= bucket_end) {                                  // the entries move on to a later bucket: flush this bucket's piece
            if (ql == 0) piece[c + b] = acc;
            acc = C::infinity();
            do { b++; bucket_end = off[b + 1]; } while (e >= bucket_end);      // (empty buckets in between have no entries, hence no pieces)
        }
        const uint32_t p = p_next;
        typename C::Affine q;
        if constexpr (PREFETCH_POINT) {
            q = q_next;
            if (e + 1 < end) {
                p_next = sorted[e + 1];
                const uint32_t idx = entry_index(p_next);
                q_next = C::aff_infinity();
                if (idx != 0xffffffffu) q_next = table[idx];
            }
        } else {
            if (e + 1 < end) p_next = sorted[e + 1];
            const uint32_t idx = entry_index(p);
            if (idx == 0xffffffffu) continue;
            q = table[idx];
        }
        if (p >> 31) q = C::neg(q);
        acc = C::template maddV<Q>(acc, q, ql);                 // an absent entry is the point at infinity: acc unchanged
    }
    if (ql == 0) piece[c + b] = acc;
}

// thread / quad per bucket: sum its chunk pieces (serial; buckets over MSM_HEAVY pieces are queued).
// The bucket-reduction tails are chains of dependent group operations run by few waves: their launch bounds ask for
// registers (no scratch), not for occupancy.
// the chunk pieces of bucket b under one sort: [s0, s1) (empty bucket: s0 == s1)
struct PieceRange { uint32_t s0, s1; ZK_HD uint32_t n() const { return s1 - s0; } };
static ZK_D PieceRange msm_piece_range(const uint32_t *__restrict__ off, uint32_t nb, const ChunkRule &rule, uint32_t b) {
    const uint32_t e0 = off[b], e1 = off[b + 1], seg = rule.len(off[nb]);
    PieceRange r; r.s0 = r.s1 = 0;
    if (e1 > e0) { r.s0 = e0 / seg + b; r.s1 = (e1 - 1) / seg + b + 1; }
    return r;
}
// piece2 != nullptr: a SECOND accumulation over the same bucket ids (another sort: off2, rule2) whose pieces are folded into the same
// bucket sums -- the H- and the L-query only ever meet as C = Ht + Lt (tcc:540), so ONE bucket reduction serves both (MsmWork::enqueue_tail)
template <class C, int Q>
__global__ void __launch_bounds__(64, C::WAVES_PER_SIMD / 2)
k_msm_bucket_finalize(const typename C::XYZZ *__restrict__ piece, const uint32_t *__restrict__ off,
                      uint32_t nb, ChunkRule rule,
                      const typename C::XYZZ *__restrict__ piece2, const uint32_t *__restrict__ off2, ChunkRule rule2,
                      typename C::XYZZ *__restrict__ bucket,
                      uint32_t *__restrict__ heavy_list, uint32_t *__restrict__ heavy_count) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, b = gt / Q, ql = gt % Q;
    if (b >= nb) return;
    const PieceRange r1 = msm_piece_range(off, nb, rule, b);
    PieceRange r2; r2.s0 = r2.s1 = 0;
    if (piece2) r2 = msm_piece_range(off2, nb, rule2, b);
    if (r1.n() + r2.n() > MSM_HEAVY) { if (ql == 0) heavy_list[atomicAdd(heavy_count, 1u)] = b; return; }
    typename C::XYZZ acc = C::infinity();
    for (uint32_t s = r1.s0; s < r1.s1; s++) acc = C::template addQ<Q>(acc, piece[s], ql);
    for (uint32_t s = r2.s0; s < r2.s1; s++) acc = C::template addQ<Q>(acc, piece2[s], ql);
    if (ql == 0) bucket[b] = acc;
}

// workgroups walk the heavy list; 128 threads / quads stride over the bucket's pieces, then an LDS tree.
// LDS holds 64 points (the upper half parks, the lower half adds) to stay inside 64 KiB for G2.
template <class C, int Q>
__global__ void __launch_bounds__(128 * Q)
k_msm_heavy(const typename C::XYZZ *__restrict__ piece, const uint32_t *__restrict__ off, uint32_t nb, ChunkRule rule,
            const typename C::XYZZ *__restrict__ piece2, const uint32_t *__restrict__ off2, ChunkRule rule2,
            const uint32_t *__restrict__ heavy_list, const uint32_t *__restrict__ heavy_count,
            typename C::XYZZ *__restrict__ bucket) {
    __shared__ typename C::XYZZ sh[64];
    const uint32_t nheavy = *heavy_count, lt = threadIdx.x / Q, ql = threadIdx.x % Q;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const uint32_t b = heavy_list[h];
        const PieceRange r1 = msm_piece_range(off, nb, rule, b);
        PieceRange r2; r2.s0 = r2.s1 = 0;
        if (piece2) r2 = msm_piece_range(off2, nb, rule2, b);
        typename C::XYZZ acc = C::infinity();
        for (uint32_t i = lt; i < r1.n() + r2.n(); i += 128)         // the two runs of pieces, one after the other
            acc = C::template addQ<Q>(acc, i < r1.n() ? piece[r1.s0 + i] : piece2[r2.s0 + (i - r1.n())], ql);
        for (uint32_t half = 64; half > 0; half >>= 1) {
            if (lt >= half && lt < 2 * half && ql == 0) sh[lt - half] = acc;
            __syncthreads();
            if (lt < half) acc = C::template addQ<Q>(acc, sh[lt], ql);
            __syncthreads();
        }
        if (threadIdx.x == 0) bucket[b] = acc;
    }
}

// ---- bucket sums -> sum_b (b + 1) B_b WITHOUT scalar multiplications (round 4).  Write bucket b = hi * L + lo (L = 2^lo_bits columns, H = 2^hi_bits rows):
//     sum_b (b + 1) B_b  =  L * sum_hi hi * Row_hi  +  sum_lo (lo + 1) * Col_lo,      Row_hi = sum_lo B_{hi,lo},  Col_lo = sum_hi B_{hi,lo}
// k_msm_rowcol_sum forms the H + L sums: 2 additions per bucket, like the running sums of the per-group reduction of rounds 1-3 (a thread per 8 buckets:
// run += B_j, acc += run, then acc += mul_small(run, 8 g)), but without that scalar multiplication per group -- 60 % of its dependent chain and 196 k of
// the 640 k point operations of a 2^16-bucket tail -- and in kernels that fit beside a resident accumulation wave (the G2 group reduction held 432
// registers per lane).  Same-box A/B at 2^20, three proofs in flight: +2.7 % on one box, +7.9 % on another; one synchronous proof 2^16 1.60 -> 1.38 ms,
// 2^18 3.38 -> 3.20, 2^20 9.97 -> 9.82; unbatched Merkle-29 785 -> 1 042 proofs/s (profiles/r04_rowcol_tail_ab.txt).
// k_msm_weighted_sum: ONE workgroup per bucket set turns the H row sums and the L column sums into the result: each half of the workgroup forms
// sum_i i X_i of its vector (a thread's own run serially, the thread totals by a suffix scan: sum_t t S_t = sum_{k >= 1} suffix_k, one LDS tree), the
// row half scales by L (lo_bits doublings), the column half adds sum_i X_i.
// `seg` logical threads (a power of two, <= T) share one row / column: each sums n / seg of its elements serially, an LDS tree of log2(seg) levels
// follows; a workgroup of T logical threads takes T / seg rows (or columns).  seg = 16 keeps 84 % of the lane-cycles busy (a 256-wide tree over 256
// elements: 11 %) -- the throughput shape; seg = T is the shortest chain -- the shape of one synchronous proof.  Rows first, then columns:
// workgroups [0, row_groups) take rows, the rest columns.
template <class C, int Q>
__global__ void __launch_bounds__(MSM_TREE)
k_msm_rowcol_sum(const typename C::XYZZ *__restrict__ bucket, uint32_t lo_bits, uint32_t hi_bits, uint32_t seg_row, uint32_t seg_col, uint32_t row_groups,
                 typename C::XYZZ *__restrict__ rc) {
    constexpr uint32_t T = MSM_TREE / Q;
    __shared__ typename C::XYZZ sh[T / 2];
    const uint32_t L = 1u << lo_bits, H = 1u << hi_bits;
    bucket += ((size_t)blockIdx.y << (lo_bits + hi_bits)); rc += (size_t)blockIdx.y * (H + L);      // blockIdx.y: the bucket set
    const uint32_t lt = threadIdx.x / Q, ql = threadIdx.x % Q;
    const bool row = blockIdx.x < row_groups;
    const uint32_t seg = row ? seg_row : seg_col, per = T / seg;                                   // threads per line, lines per workgroup
    const uint32_t line = (row ? blockIdx.x : blockIdx.x - row_groups) * per + lt / seg, k = lt % seg;
    const uint32_t n = row ? L : H, lines = row ? H : L;
    const bool live = line < lines;
    const uint32_t first = row ? line << lo_bits : line, step = row ? 1u : L;
    typename C::XYZZ acc = C::infinity();
    if (live) for (uint32_t i = k; i < n; i += seg) acc = C::template addQ<Q>(acc, bucket[first + (size_t)i * step], ql);
    for (uint32_t half = seg / 2; half > 0; half >>= 1) {                                           // tree inside every segment of `seg` threads
        if (k >= half && k < 2 * half && ql == 0) sh[(lt / seg) * (seg / 2) + (k - half)] = acc;
        __syncthreads();
        if (k < half) acc = C::template addQ<Q>(acc, sh[(lt / seg) * (seg / 2) + k], ql);
        __syncthreads();
    }
    if (live && k == 0 && ql == 0) rc[row ? line : H + line] = acc;
}

template <class C, int Q>
__global__ void __launch_bounds__(MSM_TREE)
k_msm_weighted_sum(const typename C::XYZZ *__restrict__ rc, uint32_t lo_bits, uint32_t hi_bits, typename C::XYZZ *__restrict__ out) {
    constexpr uint32_t TH = MSM_TREE / Q / 2;                 // logical threads per half: rows | columns
    __shared__ typename C::XYZZ sh[2 * TH];                   // (G2, one lane per thread: 256 x 256 B = 64 KiB)
    const uint32_t L = 1u << lo_bits, H = 1u << hi_bits;
    rc += (size_t)blockIdx.x * (H + L);                       // blockIdx.x: the bucket set
    const uint32_t lt = threadIdx.x / Q, ql = threadIdx.x % Q, half = lt / TH, t = lt % TH;
    const uint32_t n = half ? L : H;
    const typename C::XYZZ *X = rc + (half ? H : 0);
    typename C::XYZZ *mine = sh + half * TH;
    const uint32_t W = n < TH ? n : TH;                        // threads of this half that hold elements (powers of two); W differs between the halves
    const uint32_t Wmax = (L > H ? L : H) < TH ? (L > H ? L : H) : TH;       // ... the loop bounds are those of the wider half (uniform barriers)
    const uint32_t E = n / W, i0 = t * E;                     // elements per thread
    // own run: run = sum_e X_{i0+e}, acc = sum_e e X_{i0+e}
    typename C::XYZZ run = C::infinity(), acc = C::infinity();
    if (t < W) for (uint32_t e = E; e-- > 0;) {
        run = C::template addQ<Q>(run, X[i0 + e], ql);
        if (e) acc = C::template addQ<Q>(acc, run, ql);
    }
    // suffix scan of the thread totals: suf_t = sum_{t' >= t} run_t'
    typename C::XYZZ suf = run;
    if (ql == 0) mine[t] = suf;
    __syncthreads();
    for (uint32_t d = 1; d < Wmax; d <<= 1) {
        typename C::XYZZ other = C::infinity();
        if (t + d < W) other = mine[t + d];
        __syncthreads();
        if (t < W && d < W) { suf = C::template addQ<Q>(suf, other, ql); if (ql == 0) mine[t] = suf; }
        __syncthreads();
    }
    const typename C::XYZZ total = mine[0];                   // sum_i X_i of this half
    __syncthreads();
    // sum_i i X_i = sum_t (acc_t + E t run_t) = sum_t acc_t + E sum_{k >= 1} suf_k
    typename C::XYZZ v = (t && t < W) ? suf : C::infinity();
    for (uint32_t e = E; e > 1; e >>= 1) v = C::template dblQ<Q>(v, ql);
    v = C::template addQ<Q>(v, acc, ql);
    for (uint32_t h = Wmax / 2; h > 0; h >>= 1) {
        if (t >= h && t < 2 * h && ql == 0) mine[t - h] = v;
        __syncthreads();
        if (t < h) v = C::template addQ<Q>(v, mine[t], ql);
        __syncthreads();
    }
    // rows: L * sum_hi hi Row_hi;   columns: sum_lo lo Col_lo + sum_lo Col_lo
    if (t == 0) {
        if (half == 0) for (uint32_t b = 0; b < lo_bits; b++) v = C::template dblQ<Q>(v, ql);
        else v = C::template addQ<Q>(v, total, ql);
        if (ql == 0) mine[0] = v;
    }
    __syncthreads();
    if (lt == 0) {
        v = C::template addQ<Q>(v, sh[TH], ql);
        if (ql == 0) out[blockIdx.x] = v;
    }
}

// -------------------------------------------------------------------------------------------------
template <class C>
int MsmWork<C>::alloc(uint32_t n, uint32_t c, typename C::Affine *shared_table, const MsmShape *sort_like, bool sort_only, uint32_t batch, uint32_t plog) {
    max_batch = batch ? batch : 1;
    if (sort_like) sh = *sort_like; else sh.set(n ? n : 1, c, max_batch, plog);
    sh.set_slots(sh.acc_pairs && C::WAVES_PER_SIMD_PAIRS != C::WAVES_PER_SIMD ? C::WAVES_PER_SIMD_PAIRS : C::WAVES_PER_SIMD);
    const uint64_t B = max_batch, BS = (uint64_t)max_batch << sh.plog;         // proofs; bucket sets (proofs x planes)
    if ((uint64_t)sh.n * sh.rows() >= (1ull << 31)) return fail_msg(ZK_ERR_ARG, "MSM too large: n * table rows must stay below 2^31 (entry payload = table index | sign)");
    if (sh.max_entries() * B >= (1ull << 32) - 64) return fail_msg(ZK_ERR_ARG, "MSM too large: proofs * n * windows must stay below 2^32 (32-bit entry offsets)");
    if ((uint64_t)sh.nb * BS > (uint64_t)SORT_MAX_CB * SORT_MAX_FB) return fail_msg(ZK_ERR_ARG, "batch too large: proofs * planes * buckets exceeds the sort's 2^20 buckets");
    if (sort_only) { owns_table = false; table_n = n; }
    else if (shared_table) { table = shared_table; owns_table = false; table_n = n; }
    else ZK_HIP(hipMalloc(&table, sizeof(typename C::Affine) * (size_t)(n ? n : 1) * sh.rows()));
    ss.set((uint32_t)(sh.n * B), (uint32_t)(sh.nb * BS), sh.W);           // sized for a full batch (the bins of a smaller batch are re-derived per call)
    if (!sort_like) {
        ZK_HIP(hipMalloc(&pairs, sizeof(uint2) * sh.max_entries() * B));
        ZK_HIP(hipMalloc(&counts, sizeof(uint32_t) * ((size_t)SORT_MAX_CB * ss.groups + 1)));
        ZK_HIP(hipMalloc(&bin_total, sizeof(uint32_t) * (SORT_MAX_CB + 1)));
        ZK_HIP(hipMalloc(&bin_base, sizeof(uint32_t) * (SORT_MAX_CB + 1)));
        ZK_HIP(hipMalloc(&sorted, sizeof(uint32_t) * sh.max_entries() * B));
        ZK_HIP(hipMalloc(&off, sizeof(uint32_t) * (sh.nb * BS + 1)));
    }
    if (sort_only) return ZK_OK;
    const uint64_t n_pieces = sh.chunk.max_chunks(sh.max_entries() * B) + sh.nb * BS + 1;
    ZK_HIP(hipMalloc(&heavy_list, sizeof(uint32_t) * (n_pieces / MSM_HEAVY + 2)));
    ZK_HIP(hipMalloc(&heavy_count, sizeof(uint32_t)));
    ZK_HIP(hipMalloc(&pieces, sizeof(typename C::XYZZ) * n_pieces));
    ZK_HIP(hipMalloc(&bucket, sizeof(typename C::XYZZ) * sh.nb * BS));
    const uint32_t rowcol = (1u << sh.lo_bits()) + (1u << sh.hi_bits());          // row + column sums per bucket set (k_msm_rowcol_sum)
    ZK_HIP(hipMalloc(&partial_a, sizeof(typename C::XYZZ) * ((size_t)rowcol * BS + 1)));
    ZK_HIP(hipMalloc(&partial_b, sizeof(typename C::XYZZ) * (BS + 1)));                // one result per bucket set (k_msm_weighted_sum)
    ZK_HIP(hipHostMalloc(&host_result, sizeof(typename C::XYZZ) * BS, hipHostMallocDefault));
    ZK_HIP(hipEventCreate(&ev_acc0)); ZK_HIP(hipEventCreate(&ev_acc1));
    return ZK_OK;
}

template <class C>
void MsmWork<C>::release() {
    if (!owns_table) table = nullptr;
    void *dev[] = {table, pairs, counts, bin_total, bin_base, off, sorted, heavy_list, heavy_count, pieces, bucket, partial_a, partial_b};
    for (void *p : dev) if (p) hipFree(p);
    if (host_result) hipHostFree(host_result);
    if (ev_acc0) hipEventDestroy(ev_acc0);
    if (ev_acc1) hipEventDestroy(ev_acc1);
    *this = MsmWork();
}

template <class C>
int MsmWork<C>::precompute(const typename C::Affine *d_bases, uint32_t n, hipStream_t st) {
    if (n > sh.n) return fail_msg(ZK_ERR_ARG, "MSM precompute: more bases than the shape was allocated for");
    table_n = n;
    if (n) ZK_LAUNCH(k_msm_precompute<C>, zk_div_up(n, 64), 64, st, d_bases, n, sh.c << sh.plog, sh.rows(), table);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

template <class C>
int MsmWork<C>::enqueue_sort(const fe *scalars, const uint32_t *gather, uint32_t n, int canonical, hipStream_t st, uint32_t batch, uint32_t stride) {
    if (n != table_n || !sorted) return fail_msg(ZK_ERR_ARG, "MSM sort: scalar count differs from the table's base count");   // the table stride is the precompute-time n
    if (!batch || batch > max_batch) return fail_msg(ZK_ERR_ARG, "MSM sort: batch exceeds the capacity the context was created with");
    sort_batch = batch;
    const uint32_t c = sh.c, W = sh.W, plog = sh.plog, nb = (sh.nb * batch) << plog;      // buckets of the whole batch, all planes
    SortShape sq = ss; sq.resize(n * batch ? n * batch : 1, nb);   // bins and workgroups for this call's scalars
    ZK_LAUNCH_SYNC(k_sort_count<C>, sq.groups, SORT_THREADS, st, scalars, gather, n, batch, stride, canonical, c, W, plog, sq, counts);
    ZK_LAUNCH_SYNC(k_sort_colscan<C>, sq.cb, SORT_THREADS, st, counts, sq.groups, bin_total);
    ZK_LAUNCH_SYNC(k_sort_binscan<C>, 1, SORT_THREADS, st, (const uint32_t *)bin_total, sq.cb, bin_base);
    if ((uint64_t)sq.per_group * W <= SORT_STAGE)
        ZK_LAUNCH_SYNC(k_sort_partition_staged<C>, sq.groups, SORT_THREADS, st, scalars, gather, n, batch, stride, canonical, c, W, plog, sq,
                       (const uint32_t *)counts, (const uint32_t *)bin_total, (const uint32_t *)bin_base, pairs, sort_kbits);
    else
    ZK_LAUNCH_SYNC(k_sort_partition<C>, sq.groups, SORT_THREADS, st, scalars, gather, n, batch, stride, canonical, c, W, plog, sq,
                   (const uint32_t *)counts, (const uint32_t *)bin_base, pairs, sort_kbits);
    ZK_LAUNCH_SYNC(k_sort_fine<C>, sq.cb, SORT_FINE_THREADS, st, (const uint2 *)pairs, (const uint32_t *)bin_base, sq, nb, off, sorted);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// bucket reduction with Q lanes per logical thread
template <class C>
template <int Q>
int MsmWork<C>::launch_reduce(hipStream_t st, const MsmWork<C> *also) {
    const uint32_t nb = sh.nb * cur_batch;                    // buckets of all bucket sets (proofs x planes)
    const typename C::XYZZ *piece2 = also ? also->pieces : nullptr;
    const uint32_t *off2 = also ? also->cur_off : nullptr;
    const ChunkRule rule2 = also ? also->sh.chunk : sh.chunk;
    ZK_LAUNCH((k_msm_bucket_finalize<C, Q>), zk_div_up((uint64_t)nb * Q, 64), 64, st, (const typename C::XYZZ *)pieces, (const uint32_t *)cur_off, nb, sh.chunk,
              piece2, off2, rule2, bucket, heavy_list, heavy_count);
    ZK_LAUNCH_SYNC((k_msm_heavy<C, Q>), MSM_HEAVY_GRID, 128 * Q, st, (const typename C::XYZZ *)pieces, (const uint32_t *)cur_off, nb, sh.chunk,
                   piece2, off2, rule2, (const uint32_t *)heavy_list, (const uint32_t *)heavy_count, bucket);
    // row / column sums (partial_a), then one weighted sum per bucket set: the results land in partial_b
    const uint32_t lo_bits = sh.lo_bits(), hi_bits = sh.hi_bits();            // nb = 2^(c-1) = H L
    const uint32_t L = 1u << lo_bits, H = 1u << hi_bits, T = MSM_TREE / Q;
    // threads per row / column: the whole workgroup for latency-sized work (Q = 4: a synchronous proof or a small one), 16 where the tails share
    // the machine with the accumulations of other proofs (84 % of the lane-cycles useful instead of 11 %; ZK_ROWCOL_SEG sweep at 2^20, three proofs
    // in flight: 4: 107.1, 8: 109.4, 16: 111.5 / 110.8, 32: 109.1, 256: 106.1 proofs/s)
    const uint32_t want = Q == 4 ? T : sh.rowcol_seg;
    const uint32_t seg_row = want < L ? want : L, seg_col = want < H ? want : H;
    const uint32_t row_groups = zk_div_up(H, T / seg_row), col_groups = zk_div_up(L, T / seg_col);
    ZK_LAUNCH_SYNC((k_msm_rowcol_sum<C, Q>), dim3(row_groups + col_groups, cur_batch), MSM_TREE, st, (const typename C::XYZZ *)bucket, lo_bits, hi_bits,
                   seg_row, seg_col, row_groups, partial_a);
    ZK_LAUNCH_SYNC((k_msm_weighted_sum<C, Q>), cur_batch, MSM_TREE, st, (const typename C::XYZZ *)partial_a, lo_bits, hi_bits, partial_b);
    return ZK_OK;
}

// first half of the reduction: the machine-filling accumulation of the sorted entries into chunk pieces, on `st`
template <class C>
int MsmWork<C>::enqueue_accumulate(const SortView &v, hipStream_t st) {
    if (!v.sorted || !v.batch || v.batch > (max_batch << sh.plog) || v.nb != sh.nb * v.batch) return fail_msg(ZK_ERR_ARG, "MSM reduce: the driving sort has a different bucket count");   // a borrowed sort must have this MSM's buckets
    cur_batch = v.batch;                                        // bucket sets: proofs x planes
    const uint32_t nb = v.nb;                                   // buckets of all proofs of the batch
    // the entry count is only known on the device: launch for the most chunks it can give, threads past the end exit
    const uint64_t max_seg = sh.chunk.max_chunks(v.entries_bound);
    if (max_seg + nb + 1 > sh.chunk.max_chunks(sh.max_entries() * max_batch) + ((uint64_t)sh.nb * max_batch << sh.plog) + 1) return fail_msg(ZK_ERR_ARG, "MSM reduce: the driving sort has more entries than this MSM was allocated for");
    ZK_HIP(hipEventRecord(ev_acc0, st));
    bool pairs = false;
    if constexpr (C::WAVES_PER_SIMD_PAIRS != C::WAVES_PER_SIMD) pairs = sh.acc_pairs;      // (only a curve whose pairs form differs gets that kernel: G1)
    if (sh.quad_acc == 4)
        ZK_LAUNCH((k_msm_accumulate<C, 4>), zk_div_up(max_seg * 4, 64), 64, st, (const typename C::Affine *)table, v.sorted, v.off,
                  nb, sh.chunk, v.remap_src, v.remap_offset, v.remap_pos, table_n, v.remap_kbits, pieces);
    else if (pairs) {
        if constexpr (C::WAVES_PER_SIMD_PAIRS != C::WAVES_PER_SIMD)
            ZK_LAUNCH((k_msm_accumulate<C, 2>), zk_div_up(max_seg, 64), 64, st, (const typename C::Affine *)table, v.sorted, v.off,
                      nb, sh.chunk, v.remap_src, v.remap_offset, v.remap_pos, table_n, v.remap_kbits, pieces);
    } else
        ZK_LAUNCH((k_msm_accumulate<C, 1>), zk_div_up(max_seg, 64), 64, st, (const typename C::Affine *)table, v.sorted, v.off,
                  nb, sh.chunk, v.remap_src, v.remap_offset, v.remap_pos, table_n, v.remap_kbits, pieces);
    ZK_HIP(hipEventRecord(ev_acc1, st));
    cur_off = v.off;
    tail_pending = true;
    return ZK_OK;
}

// second half: chunk pieces -> bucket sums -> sum_b (b + 1) B_b -> one point per proof, on st_tail (behind this MSM's accumulation).
// also != nullptr: the chunk pieces of ANOTHER accumulation over the same bucket ids (same window bits, same batch; its accumulation
// queued earlier on the same accumulation stream) are folded into this MSM's buckets -- the result is the SUM of both multi-
// exponentiations, and `also` contributes the point at infinity from then on (its host_result is cleared here).
template <class C>
int MsmWork<C>::enqueue_tail(hipStream_t st_tail, uint32_t tail_lanes, MsmWork<C> *also) {
    if (!tail_pending) return fail_msg(ZK_ERR_ARG, "MSM tail: no accumulation is pending");
    if (also && (!also->tail_pending || also->sh.nb != sh.nb || also->sh.plog != sh.plog || also->cur_batch != cur_batch)) return fail_msg(ZK_ERR_ARG, "MSM tail: the accumulation to fold in has other buckets or another batch");
    tail_pending = false;
    hipStream_t st = st_tail;
    ZK_HIP(hipStreamWaitEvent(st, ev_acc1, 0));
    if (also) {
        also->tail_pending = false;
        ZK_HIP(hipStreamWaitEvent(st, also->ev_acc1, 0));
        memset(also->host_result, 0, sizeof(typename C::XYZZ) * ((size_t)also->max_batch << also->sh.plog));        // infinity (ZZ = 0): its share is inside this MSM's result
    }
    ZK_HIP(hipMemsetAsync(heavy_count, 0, sizeof(uint32_t), st));
    const uint32_t lanes = tail_lanes == 4 || tail_lanes == 1 ? tail_lanes : sh.quad;
    const int rc = lanes == 4 ? launch_reduce<4>(st, also) : launch_reduce<1>(st, also);
    if (rc != ZK_OK) return rc;
    typename C::XYZZ *cur = partial_b;
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(host_result, cur, sizeof(typename C::XYZZ) * cur_batch, hipMemcpyDeviceToHost, st));
    if (dev_result && sh.plog == 0) {                           // (frugal tables: the planes are folded on the host, which then writes the device copy: zkhip.cpp)
        if (cur_batch == 1) ZK_HIP(hipMemcpyAsync(dev_result, cur, sizeof(typename C::XYZZ), hipMemcpyDeviceToDevice, st));
        else ZK_HIP(hipMemcpy2DAsync(dev_result, dev_result_pitch, cur, sizeof(typename C::XYZZ), sizeof(typename C::XYZZ), cur_batch, hipMemcpyDeviceToDevice, st));
    }
    return ZK_OK;
}

template <class C>
int MsmWork<C>::enqueue_reduce(const SortView &v, hipStream_t st, hipStream_t st_tail, uint32_t tail_lanes) {
    ZK_TRY(enqueue_accumulate(v, st));
    return enqueue_tail(st_tail ? st_tail : st, tail_lanes, nullptr);
}

template <class C>
int MsmWork<C>::enqueue(const fe *scalars, const uint32_t *gather, uint32_t n, int canonical, hipStream_t st, hipStream_t st_tail, uint32_t batch, uint32_t stride) {
    ZK_TRY(enqueue_sort(scalars, gather, n, canonical, st, batch, stride));
    return enqueue_reduce(view(), st, st_tail);
}

}  // namespace zk
