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
    return 0;
}
