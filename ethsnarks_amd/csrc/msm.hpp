// msm.hpp -- multi-scalar multiplication sum_i s_i * P_i over alt_bn128 G1 / G2 on gfx950.
//
// Replaces (reference call sites; bodies in the ABSENT libff / libsnark forks):
//   kc_multi_exp_with_mixed_addition<G1|G2, Fr, multi_exp_method_BDLO12>   tcc:488-506  (A-, B-query)
//   libff::multi_exp<G1, Fr, BDLO12>                                       tcc:510-518  (H-query)
//   libff::multi_exp_with_mixed_addition<G1, Fr, BDLO12>                   tcc:522-530  (L-query)
// BDLO12 is the bucket (Pippenger) method; the group element it returns is unique, so any schedule
// of the additions gives the same proof bytes.
//
// MI355X-first layout: the query bases are static, HBM is 288 GB, so at context creation every base
// P_k is expanded to its W window multiples  T[w][k] = 2^(c*w) * P_k  (affine, W*n points: 1 GB per G1
// query at n = 2^20, c = 16).  A signed digit d of window w of scalar k then contributes sign(d) * T[w][k]
// to bucket |d| of ONE shared set of 2^(c-1) buckets: no per-window bucket sets, no window Horner,
// 16x fewer buckets to reduce, ~n*W/2^(c-1) entries per bucket (240 at n = 2^20, c = 17).
//
// Per proof (integer VALU work; random 64/128-B gathers from the table):
//   1 k_sort_count / k_sort_colscan / k_sort_binscan / k_sort_partition   scalar -> signed c-bit digits ->
//                       entries partitioned into <= 256 coarse bucket bins (LDS ranking, no global atomics)
//   2 k_sort_fine       one workgroup per coarse bin: LDS histogram + in-L2 scatter by bucket
//   3 k_msm_accumulate  one thread per *chunk* of consecutive sorted entries (equal length for every thread, bucket
//                       boundaries crossed inside the loop): XYZZ mixed additions   <- dominant kernel
//   4 k_msm_bucket_finalize / k_msm_heavy   chunk pieces -> bucket sums
//   5 k_msm_rowcol_sum  row and column sums of the bucket matrix (b = hi L + lo): LDS trees, no scalar multiplications
//   6 k_msm_weighted_sum  L sum_hi hi Row_hi + sum_lo (lo + 1) Col_lo = sum_b (b + 1) B_b -> the MSM result (one XYZZ point per bucket set)
// Kernels 3-6 are templated on the lanes per logical thread: <C, 1> for machine-filling sizes, <C, 4> (four lanes
// share one point operation, Curve::add_q / dbl_q / madd_q in bn254.hpp) where a proof is a chain of dependent
// additions rather than a throughput problem (MsmShape::quad / quad_acc).
// Zero scalars produce no entries; scalar 1 (and any repeated value) lands in one bucket whose entries
// are cut into chunk pieces, so 0/1-heavy witnesses (the *_with_mixed_addition fast paths of the reference)
// stay load-balanced without special cases; buckets with very many chunk pieces go to a workgroup reducer.
#pragma once
#include "bn254.hpp"
#include "common.hpp"
#include <stdlib.h>

namespace zk {

constexpr uint32_t MSM_SEG_MIN = 32;      // entries per accumulation thread (lower bound) when the entries fill the machine
constexpr uint32_t MSM_SEG_MIN_SMALL = 8; // ... and for MSMs of <= 2^23 entries, whose time is the serial chain per thread, not throughput
constexpr uint32_t MSM_SEG_MAX = 48;      // ... upper bound: above it the entries are dealt in more than one round of the machine (round 4, with the row / column
                                          // reduction: ZK_SEG_MAX 32 / 40 / 48 / 56 / 64 = 114.5 / 114.6 / 114.6 / 114.1 / 113.6 proofs/s at 2^20, two runs each --
                                          // the G2 accumulation deals its 15.7 M entries in three rounds of 40 instead of two of 60)
constexpr uint32_t MSM_HEAVY = 64;        // buckets with more chunk pieces than this are reduced by a workgroup
constexpr uint32_t MSM_TREE = 256;        // workgroup size of the row / column and weighted sums of the bucket reduction
constexpr uint32_t MSM_HEAVY_GRID = 256;  // workgroups that walk the heavy-bucket list

// shapes of the two-pass bucket sort (msm_impl.hpp); the CPU emulation of tests/emul runs these same shapes
constexpr uint32_t SORT_THREADS = 256;
constexpr uint32_t SORT_FINE_THREADS = 1024;
constexpr uint32_t SORT_MAX_CB = 256;          // coarse bins
constexpr uint32_t SORT_MAX_FB = 4096;         // fine buckets per coarse bin held in LDS (c <= 20)
constexpr uint32_t SORT_STAGE = 7680;          // (bucket, payload) pairs a partition workgroup stages in LDS (60 KiB): 512 scalars x 15 windows
static_assert(SORT_MAX_CB == SORT_THREADS, "k_sort_partition_staged: one coarse bin per thread");

struct SortShape {
    uint32_t cb, fb, fine_bits, groups, per_group;   // groups = workgroups of pass 1, per_group = scalars each
    void set(uint32_t n, uint32_t nb, uint32_t W = 15) {   // context creation: reads the tuning override once
        // scalars per workgroup: as many as the staged partition holds in LDS (SORT_STAGE pairs = per_group x W windows), a power of two, at most 512
        // (512 at W = 15; 1024 with the direct scatter measured the same times)
        per_group = 512;
        while (per_group > 32 && (uint64_t)per_group * W > SORT_STAGE) per_group >>= 1;
        if (const char *e = getenv("ZK_SORT_PER_GROUP")) { int v = atoi(e); if (v >= 64) per_group = (uint32_t)v; }   // tuning aid
        resize(n, nb);
    }
    void resize(uint32_t n, uint32_t nb) {            // same per_group for another scalar / bucket count (proving path: no getenv)
        fine_bits = 0; while (((nb + (1u << fine_bits) - 1) >> fine_bits) > SORT_MAX_CB) fine_bits++;
        fb = 1u << fine_bits;                               // fine buckets per coarse bin (a power of two)
        cb = (nb + fb - 1) >> fine_bits;                    // coarse bins; the last one may be partly filled (batches that are not a power of two)
        groups = (n + per_group - 1) / per_group; if (!groups) groups = 1;
    }
};

// Entries per accumulation thread.  The sorted entry list is cut into chunks of ONE length, whatever the bucket
// boundaries (a thread flushes its running sum when it crosses one), so every lane of every wave does the same number
// of mixed additions; the length is the smallest one that deals all entries in whole rounds of the machine
// (`slots` = accumulation threads resident at once), clamped to [seg_min, seg_max].  Device and host agree on it
// through this one function of the entry count, which only the device knows (off[nb]).
struct ChunkRule {
    uint32_t slots = 1, seg_min = MSM_SEG_MIN, seg_max = MSM_SEG_MAX;
    ZK_HD uint32_t len(uint32_t total) const {
        if ((uint64_t)seg_min * slots >= total) return seg_min;
        const uint64_t per_round = (uint64_t)seg_max * slots;
        const uint64_t rounds = (total + per_round - 1) / per_round, lanes = rounds * slots;
        const uint32_t seg = (uint32_t)((total + lanes - 1) / lanes);
        return seg < seg_min ? seg_min : seg;
    }
    // most chunks any entry count <= entries_bound can give
    uint64_t max_chunks(uint64_t entries_bound) const { return entries_bound / seg_max + slots + 1; }
};
// threads of one accumulation launch that the device holds at once (waves/SIMD x 4 SIMDs x CUs x 64 lanes)
inline uint32_t msm_machine_threads(uint32_t waves_per_simd) {
#ifdef ZK_EMUL
    return 32 * waves_per_simd;                 // the emulator has no machine to fill: small, so that tests see several rounds
#else
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return (uint32_t)cus * 4u * waves_per_simd * 64u;
#endif
}

struct MsmShape {
    uint32_t n = 0, c = 0, W = 0, nb = 0;
    // Memory-frugal tables (a key whose W-fold window-multiple tables do not fit the device): the table keeps every S-th window only,
    // T[r][k] = 2^(c S r) P_k for r < rows() = ceil(W / S), S = 2^plog.  Window w = S r + j then adds its digit times T[r][k] into bucket
    // set ("plane") j, and the multi-exponentiation is sum_j 2^(c j) R_j over the S plane results (MsmWork::finish: c doublings per plane on
    // the host).  The planes lie side by side exactly like the bucket sets of a batch: set = proof * S + j, and every kernel sees
    // batch * S sets.  S = 1 is the default layout (one plane, W rows).
    uint32_t plog = 0;
    ZK_HD uint32_t planes() const { return 1u << plog; }
    ZK_HD uint32_t rows() const { return (W + planes() - 1) >> plog; }
    // the bucket matrix of the reduction: bucket b = hi * 2^lo_bits + lo (k_msm_rowcol_sum)
    ZK_HD uint32_t lo_bits() const { return c / 2; }
    ZK_HD uint32_t hi_bits() const { return (c - 1) - c / 2; }
    ChunkRule chunk;
    uint32_t quad = 1;      // lanes per logical thread in the bucket-reduction kernels: 1, or 4 (Curve::*_q) for latency-bound sizes
    uint32_t quad_acc = 1;  // ... and in the accumulation kernel (only while 4 lanes per chunk still fit the machine at once)
    bool acc_pairs = false; // G1 accumulation with dual-issue product pairs at 3 waves/SIMD (k_msm_accumulate<C, 2>): machine-filling sizes only
    uint32_t rowcol_seg = 16; // threads that share one row / column sum in k_msm_rowcol_sum when the tail is throughput work (one lane per thread)
    // largest window with >= T entries per bucket on average (n * W entries over 2^(c-1) buckets).  T = 32 for one proof at a time:
    // the bucket reduction is latency there and a larger window shortens the accumulation chains.  A batch pays the reduction of
    // every proof's buckets in throughput once its entries fill the machine, and wants fuller buckets: T grows with the entries of
    // the whole batch, 32 at 2^20.5 of them to 100 from 2^22 on (measured, proofs/s with the window this rule picks against the
    // single-proof one: MiMC-11 k = 32 8.6k -> 13.3k, k = 64 10.1k -> 16.0k; Merkle-29 k = 8 2.62k -> 2.74k, k = 32 3.58k -> 3.75k;
    // MiMC-11 k <= 16 and every single proof keep the larger window)
    static uint32_t pick_c(uint32_t n, uint32_t batch = 1) {
        for (uint32_t c = 17; c > 2; c--) {    // 17: measured optimum at n = 2^20 (W = 15 windows, 2^16 buckets)
            const uint64_t entries = (uint64_t)n * (254 / c + 1);
            const uint64_t floor_c = c >= 16 ? 2 * MSM_SEG_MIN - 1 : MSM_SEG_MIN;   // 2^15 buckets and more: 64 entries each (63: a domain's m - 1
                                                    // H scalars must not fall one short of what its m + 1 witness entries reach) -- measured with three
                                                    // proofs in flight: 2^16 constraints 591 -> 691 proofs/s with c = 15 instead of 16, 2^18 306 -> 312 with 16
                                                    // instead of 17; 2^14, 2^15, 2^17, 2^19, 2^20 keep their windows)
            uint64_t per_bucket = floor_c;
            if (batch > 1) {
                per_bucket = 32 * entries * batch / 1482910;           // 2^20.5
                if (per_bucket > 100) per_bucket = 100;
                if (per_bucket < floor_c) per_bucket = floor_c;
            }
            if (entries >= (per_bucket << (c - 1))) return c;
        }
        return 2;
    }
    // batch: proofs per launch sequence the MSM will see (the latency / throughput switches below look at all their entries)
    void set(uint32_t n_, uint32_t c_, uint32_t batch = 1, uint32_t plog_ = 0) {
        plog = plog_;
        n = n_; c = c_ ? c_ : pick_c(n_, batch ? batch : 1);
        if (c < 2) c = 2;
        if (c > 20) c = 20;
        W = 254 / c + 1; nb = 1u << (c - 1);
        // measured (tools/dev_small_sweep.sh, domains 2^13 .. 2^19): quad reductions win up to 2^17 constraints with one or
        // three proofs in flight (and still for a single proof beyond); quad accumulation only while the GPU is mostly idle
        const uint64_t all = max_entries() * (batch ? batch : 1);
        quad = all <= (1ull << 21) ? 4 : 1;
        quad_acc = all <= (1ull << 17) ? 4 : 1;
        if (const char *e = getenv("ZK_MSM_QUAD")) quad = atoi(e) ? 4 : 1;                                           // tuning aids
        if (const char *e = getenv("ZK_MSM_QUAD_ACC")) quad_acc = atoi(e) ? 4 : 1;
        // the pairs form wins where the entries fill the machine several times over; below, fewer resident threads only lengthen the chunks
        // (synchronous proofs, tools/dev_sync_latency.py, plain -> pairs: 2^17 2.38 -> 2.59 ms, 2^18 3.56 -> 3.72, 2^20 10.8 -> 10.4)
        acc_pairs = all >= (3ull << 21);
        if (const char *e = getenv("ZK_ACC_PAIRS")) acc_pairs = atoi(e) != 0;
        if (const char *e = getenv("ZK_ROWCOL_SEG")) { int v = atoi(e); if (v >= 1 && v <= 256 && !(v & (v - 1))) rowcol_seg = (uint32_t)v; }
        chunk.seg_min = all <= (1ull << 23) ? MSM_SEG_MIN_SMALL : MSM_SEG_MIN;
        chunk.seg_max = MSM_SEG_MAX;
        if (const char *e = getenv("ZK_SEG_MIN")) { int v = atoi(e); if (v >= 4) chunk.seg_min = (uint32_t)v; }    // tuning aids
        if (const char *e = getenv("ZK_SEG_MAX")) { int v = atoi(e); if (v >= 4) chunk.seg_max = (uint32_t)v; }
        if (chunk.seg_max < chunk.seg_min) chunk.seg_max = chunk.seg_min;
    }
    // the slot count depends on the curve (registers per thread): MsmWork<C>::alloc completes the rule
    void set_slots(uint32_t waves_per_simd) { chunk.slots = msm_machine_threads(waves_per_simd) / quad_acc; if (!chunk.slots) chunk.slots = 1; }
    uint64_t max_entries() const { return (uint64_t)n * W; }
    uint64_t max_chunks() const { return chunk.max_chunks(max_entries()); }
    uint64_t max_pieces() const { return max_chunks() + nb + 1; }       // a chunk starts one piece, every bucket boundary one more
};

// what k_msm_accumulate needs from a finished bucket sort; `remap_src` != 0 means the entries index another
// MSM's table ([w][remap_src] layout, scalar i) and this MSM covers scalars [remap_offset, remap_offset + n)
struct SortView {
    const uint32_t *sorted = nullptr, *off = nullptr;   // off[b] = entries before bucket b, off[nb] = all entries
    uint32_t nb = 0, entries_bound = 0;             // nb: buckets of the sort (sets x per-set buckets); entries_bound: upper bound of sorted entries (batch x n_src x W)
    uint32_t batch = 1;                             // bucket SETS sorted together (proofs x planes, MsmShape::plog): bucket id = set * (nb / sets) + digit bucket
    uint32_t remap_src = 0, remap_offset = 0;
    uint32_t remap_kbits = 0;                       // != 0: the entries are (window << kbits) | scalar instead of window * remap_src + scalar
    const uint32_t *remap_pos = nullptr;            // optional scalar index -> own base index (0xffffffff: absent); else i - remap_offset
};

// Several proofs of ONE circuit through one launch sequence: `batch` scalar vectors over the same bases.
// The sort key becomes (proof, bucket) -- bucket id = proof * nb + digit bucket, so the k bucket sets lie side by side --
// and every kernel simply sees k times the entries, buckets, groups and partial sums; the table is shared.
template <class C>
struct MsmWork {
    MsmShape sh;
    SortShape ss;
    uint32_t max_batch = 1;                     // capacity: proofs per launch sequence the buffers are sized for
    uint32_t cur_batch = 1;                     // bucket SETS (proofs x planes) of the reduction in flight
    typename C::Affine *table = nullptr;        // [W][table_n] window multiples of the bases, resident for the context's life
    uint32_t table_n = 0;
    bool owns_table = true;                     // false: the table belongs to a DeviceTables entry shared by several contexts
    uint2 *pairs = nullptr;                     // pass-1 output of the bucket sort: (payload, bucket)
    uint32_t *counts = nullptr, *bin_total = nullptr, *bin_base = nullptr;
    uint32_t *off = nullptr, *sorted = nullptr;
    uint32_t *heavy_list = nullptr, *heavy_count = nullptr;
    typename C::XYZZ *pieces = nullptr, *bucket = nullptr, *partial_a = nullptr, *partial_b = nullptr;
    typename C::XYZZ *host_result = nullptr;    // pinned, max_batch x planes entries
    typename C::XYZZ *dev_result = nullptr;     // optional: a device copy of the results as well, one every dev_result_pitch bytes (sharded provers exchange it with RCCL)
    size_t dev_result_pitch = 0;
    hipEvent_t ev_acc0 = nullptr, ev_acc1 = nullptr;   // bracket k_msm_accumulate (the dominant kernel) on its stream
    float accumulate_ms() const { float ms = 0; if (ev_acc0 && ev_acc1) hipEventElapsedTime(&ms, ev_acc0, ev_acc1); return ms; }

    // shared_table: an already expanded table of exactly these n bases (same c) to borrow; nullptr: allocate one.
    // sort_like: this MSM will be driven by ANOTHER MsmWork's sort (same scalars: A-, B- and L-query all read the
    // witness): take that shape (same c, W, buckets, chunk rule), size the reduction buffers for its entries,
    // and do not allocate sort buffers of its own.
    // sort_only: no table and no reduction buffers -- this object only sorts a scalar vector for others.
    int alloc(uint32_t n, uint32_t c, typename C::Affine *shared_table = nullptr, const MsmShape *sort_like = nullptr, bool sort_only = false, uint32_t batch = 1, uint32_t plog = 0);
    void release();
    // table <- window multiples of d_bases[0..n) (device pointer); once per context
    int precompute(const typename C::Affine *d_bases, uint32_t n, hipStream_t st);
    // enqueue the MSM: sort + accumulation (machine-filling) on `st`, the low-parallelism bucket reduction on
    // `st_tail` (may equal st); the result lands in host_result after st_tail drains
    // batch > 1: proof p reads scalars[p * stride + (gather ? gather[i] : i)]
    int enqueue(const fe *scalars, const uint32_t *gather, uint32_t n, int canonical, hipStream_t st, hipStream_t st_tail, uint32_t batch = 1, uint32_t stride = 0);
    // the two halves: the bucket sort of this MSM's scalars, and accumulation + reduction driven by a sort view
    int enqueue_sort(const fe *scalars, const uint32_t *gather, uint32_t n, int canonical, hipStream_t st, uint32_t batch = 1, uint32_t stride = 0);
    // tail_lanes: lanes per logical thread of the bucket-reduction kernels for this call (0 = the shape's choice, sh.quad)
    int enqueue_reduce(const SortView &v, hipStream_t st, hipStream_t st_tail, uint32_t tail_lanes = 0);
    // ... and enqueue_reduce's own halves: the accumulation (chunk pieces) on st, the bucket reduction on st_tail.  `also`: another
    // MsmWork whose pending chunk pieces (same buckets, same batch) are folded into THIS reduction, so two multi-exponentiations
    // whose results are only ever added -- C = Ht + Lt, tcc:540 -- share one tail
    int enqueue_accumulate(const SortView &v, hipStream_t st);
    int enqueue_tail(hipStream_t st_tail, uint32_t tail_lanes = 0, MsmWork *also = nullptr);
    bool tail_pending = false;                  // an accumulation whose chunk pieces no tail has consumed yet
    template <int Q> int launch_reduce(hipStream_t st, const MsmWork *also);
    const uint32_t *cur_off = nullptr;          // bucket offsets of the sort driving the current reduction
    uint32_t sort_batch = 1;                    // batch of the last enqueue_sort
    uint32_t sort_kbits = 0;                    // sort-only objects (the shared witness sort): entries carry (window << kbits) | scalar -- set by use_shift_payload()
    void use_shift_payload() { uint32_t k = 1; while ((1ull << k) < table_n) k++; if (((uint64_t)sh.rows() << k) < (1ull << 31)) sort_kbits = k; }
    SortView view() const { SortView v; v.sorted = sorted; v.off = off; v.batch = sort_batch << sh.plog; v.nb = sh.nb * v.batch; v.entries_bound = (uint32_t)((uint64_t)table_n * sh.W * sort_batch); return v; }
    // view for an MSM over scalars [offset, offset + n_dst) of THIS sort (its table has stride n_dst)
    SortView view_for(uint32_t offset, const uint32_t *pos = nullptr) const { SortView v = view(); v.remap_src = table_n; v.remap_offset = offset; v.remap_pos = pos; v.remap_kbits = sort_kbits; return v; }
    // device values are loose ([0, 2p)): normalise once.  Frugal tables: sum_j 2^(c j) R_j over the planes of the proof, Horner from the top plane
    typename C::XYZZ finish(uint32_t proof = 0) const {
        const uint32_t S = sh.planes();
        typename C::XYZZ r = C::canon(host_result[(size_t)proof * S + (S - 1)]);
        for (uint32_t j = S - 1; j-- > 0;) {
            for (uint32_t d = 0; d < sh.c; d++) r = C::dbl(r);
            r = C::add(r, C::canon(host_result[(size_t)proof * S + j]));
        }
        return S > 1 ? C::canon(r) : r;
    }
};

}  // namespace zk
