// tools/affine_bench.cpp -- measured answer to "would batched-affine bucket accumulation beat the XYZZ mixed addition?"
// (VERDICT r1, next-5 (iv)).  An affine addition with a shared inversion (Montgomery's trick) costs 6 field products
// instead of the ~9.5 of an XYZZ mixed addition, but the inversion (~380 products, every lane of the wave pays it) must be
// amortised over a batch of B INDEPENDENT additions per thread, whose B prefix products cannot live in registers or LDS
// (B x 32 B per thread) and travel through global memory, as do the operands a second time.
// Kernel: every thread adds B pairs (P_i, Q_i) gathered from an L2-resident table of affine G1 points, prefix products in
// a coalesced [i][thread] scratch array; results stored.  Compared with the XYZZ mixed-addition chain of tools/mulbench.cpp
// on the same table (operands off the curve are fine for timing: the formulas are polynomial identities).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ethsnarks_amd/csrc tools/affine_bench.cpp -o tools/affine_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "bn254.hpp"
namespace zk { thread_local char g_last_error[256] = ""; }
using namespace zk;

__device__ __forceinline__ fe fq_inv_device(const fe &a) {          // Fermat, a^(q-2): 254 squarings + one product per set bit
    fe acc = Fq::one(), base = a;
#pragma unroll 1
    for (int i = 0; i < 254; i++) {
        uint32_t e = FqParams::p(i >> 5);
        if ((i >> 5) == 0) e -= 2;
        if ((e >> (i & 31)) & 1) acc = Fq::lmul(acc, base);
        base = Fq::lsqr(base);
    }
    return acc;
}

template <int WPS>
__global__ void __launch_bounds__(64, WPS)
k_affine_batch(const G1::Affine *__restrict__ pts, uint32_t npts, fe *__restrict__ prefix, G1::Affine *__restrict__ out, int B) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
    fe acc = Fq::one();
    uint32_t j = t * 7 + 1;
    for (int i = 0; i < B; i++) {                                   // pass 1: denominators and their prefix products
        const fe xp = pts[j % npts].x, xq = pts[(j + 5) % npts].x;
        prefix[(size_t)i * T + t] = acc;
        acc = Fq::lmul(acc, Fq::lsub(xq, xp));
        j += 13;
    }
    acc = fq_inv_device(acc);
    for (int i = B - 1; i >= 0; i--) {                              // pass 2: unwind, one affine addition per pair
        j -= 13;
        const G1::Affine p = pts[j % npts], q = pts[(j + 5) % npts];
        const fe d = Fq::lsub(q.x, p.x);
        const fe inv = Fq::lmul(acc, prefix[(size_t)i * T + t]);
        acc = Fq::lmul(acc, d);
        const fe lam = Fq::lmul(Fq::lsub(q.y, p.y), inv);
        G1::Affine r;
        r.x = Fq::lsub(Fq::lsub(Fq::lsqr(lam), p.x), q.x);
        r.y = Fq::lsub(Fq::lmul(lam, Fq::lsub(p.x, r.x)), p.y);
        out[(size_t)i * T + t] = r;
    }
}

template <int WPS>
__global__ void __launch_bounds__(64, WPS)
k_xyzz_chain(const G1::Affine *__restrict__ pts, uint32_t npts, G1::XYZZ *__restrict__ out, int B) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    G1::XYZZ acc = G1::from_affine(pts[t % npts]);
    uint32_t j = t * 7 + 1;
    for (int i = 0; i < B; i++) { acc = G1::madd(acc, pts[j % npts]); j += 13; }
    out[t] = acc;
}

// ---- the same question over Fq2 (round 4): an affine G2 addition with a shared inversion is 16.7 Fq-product units (prefix product, two unwinding
// products, lambda, lambda^2 as a complex squaring, y3) + (380 + ~10) / B for the inversion (one Fq inversion of the norm), against 26.5 for the XYZZ
// mixed addition -- on paper -26 % at B = 128, where over Fq the trade is a wash.  Prefix products are 64 B, results 128 B.
__device__ __forceinline__ fe2 fq2_inv_device(const fe2 &a) {
    const fe n = fq_inv_device(Fq::ladd(Fq::lsqr(a.c0), Fq::lsqr(a.c1)));
    fe2 r; r.c0 = Fq::lmul(a.c0, n); r.c1 = Fq::lneg(Fq::lmul(a.c1, n)); return r;
}
// (gather indices are hashed: random 128-byte reads like the accumulation kernel's, not a strided walk)
static __device__ __forceinline__ uint32_t gidx(uint32_t j, uint32_t npts) { return (j * 2654435761u) & (npts - 1); }
template <int WPS>
__global__ void __launch_bounds__(64, WPS)
k_affine_batch_g2(const G2::Affine *__restrict__ pts, uint32_t npts, fe2 *__restrict__ prefix, G2::Affine *__restrict__ out, int B) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
    fe2 acc = Fq2::one();
    uint32_t j = t * 7 + 1;
    for (int i = 0; i < B; i++) {
        const fe2 xp = pts[gidx(j, npts)].x, xq = pts[gidx(j + 5, npts)].x;
        prefix[(size_t)i * T + t] = acc;
        acc = Fq2::lmul(acc, Fq2::lsub(xq, xp));
        j += 13;
    }
    acc = fq2_inv_device(acc);
    for (int i = B - 1; i >= 0; i--) {
        j -= 13;
        const G2::Affine p = pts[gidx(j, npts)], q = pts[gidx(j + 5, npts)];
        const fe2 d = Fq2::lsub(q.x, p.x);
        const fe2 inv = Fq2::lmul(acc, prefix[(size_t)i * T + t]);
        acc = Fq2::lmul(acc, d);
        const fe2 lam = Fq2::lmul(Fq2::lsub(q.y, p.y), inv);
        G2::Affine r;
        r.x = Fq2::lsub(Fq2::lsub(Fq2::lsqr(lam), p.x), q.x);
        r.y = Fq2::lsub(Fq2::lmul(lam, Fq2::lsub(p.x, r.x)), p.y);
        out[(size_t)i * T + t] = r;
    }
}
template <int WPS>
__global__ void __launch_bounds__(64, WPS)
k_xyzz_chain_g2(const G2::Affine *__restrict__ pts, uint32_t npts, G2::XYZZ *__restrict__ out, int B) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    G2::XYZZ acc = G2::from_affine(pts[t % npts]);
    uint32_t j = t * 7 + 1;
    for (int i = 0; i < B; i++) { acc = G2::madd(acc, pts[gidx(j, npts)]); j += 13; }
    out[t] = acc;
}

template <class F> float best_of(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}

int main() {
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int CU = p.multiProcessorCount;
    const uint32_t npts = 1u << 16;
    std::vector<uint32_t> hp(16 * (size_t)npts);
    for (size_t i = 0; i < hp.size(); i++) hp[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
    G1::Affine *pts; hipMalloc(&pts, 64 * (size_t)npts); hipMemcpy(pts, hp.data(), 64 * (size_t)npts, hipMemcpyHostToDevice);
    const int threads = CU * 4 * 4 * 64;                            // 4 waves per SIMD
    const int Bmax = 512;
    fe *prefix; G1::Affine *out; G1::XYZZ *xo;
    hipMalloc(&prefix, 32 * (size_t)threads * Bmax); hipMalloc(&out, 64 * (size_t)threads * Bmax); hipMalloc(&xo, 128 * (size_t)threads);
    for (int B : {32, 128, 512}) {
        float a = best_of([&] { hipLaunchKernelGGL(k_affine_batch<4>, dim3(threads / 64), dim3(64), 0, 0, (const G1::Affine *)pts, npts, prefix, out, B); });
        float x = best_of([&] { hipLaunchKernelGGL(k_xyzz_chain<4>, dim3(threads / 64), dim3(64), 0, 0, (const G1::Affine *)pts, npts, xo, B); });
        printf("B = %3d additions per thread, %d threads: batched affine %8.3f ms = %6.2f G add/s (prefix + result traffic %.2f GB)   |   XYZZ mixed addition chain %8.3f ms = %6.2f G add/s\n",
               B, threads, a, (double)threads * B / a * 1e-6, (double)threads * B * (32 * 2 + 64) * 1e-9, x, (double)threads * B / x * 1e-6);
    }
    hipFree(prefix); hipFree(out); hipFree(xo); hipFree(pts);
    // ---- G2: gathers from an L2-resident table (2^16 points) and from a 2 GB one (2^24 points: what the accumulation kernel reads), 2 waves/SIMD
    for (uint32_t logn : {16u, 24u}) {
        const uint32_t n2 = 1u << logn;
        G2::Affine *p2; if (hipMalloc(&p2, 128 * (size_t)n2) != hipSuccess) { printf("no memory for 2^%u G2 points\n", logn); break; }
        {
            std::vector<uint32_t> h2(32 * (size_t)(1u << 16));
            for (size_t i = 0; i < h2.size(); i++) h2[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
            for (size_t o = 0; o < n2; o += 1u << 16) hipMemcpy((char *)p2 + 128 * o, h2.data(), 128 * (size_t)(1u << 16), hipMemcpyHostToDevice);
        }
        const int th2 = CU * 4 * 2 * 64;                           // 2 waves per SIMD
        fe2 *pre2; G2::Affine *out2; G2::XYZZ *xo2;
        hipMalloc(&pre2, 64 * (size_t)th2 * 256 * 3 / 2); hipMalloc(&out2, 128 * (size_t)th2 * 256 * 3 / 2); hipMalloc(&xo2, 256 * (size_t)th2);
        for (int B : {32, 64, 128, 256}) {
            float a = best_of([&] { hipLaunchKernelGGL(k_affine_batch_g2<2>, dim3(th2 / 64), dim3(64), 0, 0, (const G2::Affine *)p2, n2, pre2, out2, B); });
            float x = best_of([&] { hipLaunchKernelGGL(k_xyzz_chain_g2<2>, dim3(th2 / 64), dim3(64), 0, 0, (const G2::Affine *)p2, n2, xo2, B); });
            float a3 = best_of([&] { hipLaunchKernelGGL(k_affine_batch_g2<3>, dim3(th2 * 3 / 2 / 64), dim3(64), 0, 0, (const G2::Affine *)p2, n2, pre2, out2, B); });
            printf("G2, table of 2^%u points, B = %3d additions per thread, %d threads: batched affine %8.3f ms = %6.3f G add/s (prefix + result traffic %.2f GB; at 3 waves/SIMD, %d threads: %6.3f G add/s)   |   XYZZ mixed addition chain %8.3f ms = %6.3f G add/s\n",
                   logn, B, th2, a, (double)th2 * B / a * 1e-6, (double)th2 * B * (64 * 2 + 128) * 1e-9, th2 * 3 / 2, (double)th2 * 1.5 * B / a3 * 1e-6, x, (double)th2 * B / x * 1e-6);
        }
        hipFree(pre2); hipFree(out2); hipFree(xo2); hipFree(p2);
    }
    return 0;
}
