// tools/mulbench.cpp -- throughput of the device field multiplication and of the XYZZ mixed addition chains at the
// occupancies the accumulation kernels run at; one binary per arithmetic variant (-DZK_FIPS_V1, -DZK_M_MULLO, ...).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ethsnarks_amd/csrc tools/mulbench.cpp -o tools/mulbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "bn254.hpp"
namespace zk { thread_local char g_last_error[256] = ""; }
using namespace zk;

template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_mul1(fe *io, int iters) {          // one dependent chain per thread
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x];
    for (int i = 0; i < iters; i++) { a = Fq::lmul(a, b); b = Fq::lmul(b, a); }
    io[t] = Fq::ladd(a, b);
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_mul2(fe *io, int iters) {          // two independent chains per thread
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x], c = Fq::ladd(a, b), d = Fq::ladd(b, b);
    for (int i = 0; i < iters; i++) { a = Fq::lmul(a, b); c = Fq::lmul(c, d); b = Fq::lmul(b, a); d = Fq::lmul(d, c); }
    io[t] = Fq::ladd(Fq::ladd(a, b), Fq::ladd(c, d));
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_mul2x(fe *io, int iters) {         // two independent chains per thread, dual-issue statements (lmul_x2)
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x], c = Fq::ladd(a, b), d = Fq::ladd(b, b);
    for (int i = 0; i < iters; i++) { Fq::lmul_x2(a, b, c, d, a, c); Fq::lmul_x2(b, a, d, c, b, d); }
    io[t] = Fq::ladd(Fq::ladd(a, b), Fq::ladd(c, d));
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_dot2(fe *io, int iters) {          // lmul2 chain (what an Fq2 product is made of)
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x], c = Fq::ladd(a, b), d = Fq::ladd(b, b);
    for (int i = 0; i < iters; i++) { a = Fq::lmul2(a, b, c, d); c = Fq::lmul2(c, d, a, b); }
    io[t] = Fq::ladd(a, c);
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_fq2mul(fe *io, int iters) {        // Fq2 product chain: two independent dot products per product
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe2 x, y; x.c0 = io[t]; x.c1 = io[t + gridDim.x * blockDim.x]; y.c0 = Fq::ladd(x.c0, x.c1); y.c1 = Fq::ladd(x.c1, x.c1);
    for (int i = 0; i < iters; i++) { x = Fq2::lmul(x, y); y = Fq2::lmul(y, x); }
    io[t] = Fq::ladd(Fq::ladd(x.c0, x.c1), Fq::ladd(y.c0, y.c1));
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_dot2x2(fe *io, int iters) {        // two independent lmul2 chains per thread
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x], c = Fq::ladd(a, b), d = Fq::ladd(b, b);
    fe e = Fq::ladd(a, c), f = Fq::ladd(b, d), g = Fq::ladd(c, d), h = Fq::ladd(d, d);
    for (int i = 0; i < iters; i++) { a = Fq::lmul2(a, b, c, d); e = Fq::lmul2(e, f, g, h); c = Fq::lmul2(c, d, a, b); g = Fq::lmul2(g, h, e, f); }
    io[t] = Fq::ladd(Fq::ladd(a, c), Fq::ladd(e, g));
}
template <class C, int WPS>
__global__ void __launch_bounds__(64, WPS) k_madd(const typename C::Affine *pts, typename C::XYZZ *out, int iters, uint32_t npts) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    typename C::XYZZ acc = C::from_affine(pts[t % npts]);
    uint32_t j = t * 7 + 1;
    for (int i = 0; i < iters; i++) { acc = C::madd(acc, pts[j % npts]); j += 13; }   // operands off the curve are fine for timing
    out[t] = acc;
}

template <class K, class... A>
float time_kernel(K k, int grid, int block, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, args...);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, args...);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const char *tag = argc > 1 ? argv[1] : "";
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int CU = p.multiProcessorCount, block = 256, it = 400;
    const int maxg = CU * 8;
    fe *io; hipMalloc(&io, 32 * 2 * (size_t)maxg * block);
    std::vector<uint32_t> h(16 * (size_t)maxg * block);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
    hipMemcpy(io, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
#define RUN(name, kern, wps, mulsper) { int g = CU * wps; float ms = time_kernel(kern<wps>, g, block, io, it); \
        printf("%-10s %-14s %d waves/SIMD: %8.3f ms  %7.2f G Fq-mul-equiv/s\n", tag, name, wps, ms, (double)mulsper * g * block * it / ms * 1e-6); }
    RUN("mul chain", k_mul1, 1, 2) RUN("mul chain", k_mul1, 2, 2) RUN("mul chain", k_mul1, 4, 2) RUN("mul chain", k_mul1, 8, 2)
    RUN("mul 2-ilp", k_mul2, 1, 4) RUN("mul 2-ilp", k_mul2, 2, 4) RUN("mul 2-ilp", k_mul2, 4, 4)
    RUN("mul x2", k_mul2x, 1, 4) RUN("mul x2", k_mul2x, 2, 4) RUN("mul x2", k_mul2x, 3, 4) RUN("mul x2", k_mul2x, 4, 4)
    RUN("dot2 chain", k_dot2, 2, 2) RUN("dot2 chain", k_dot2, 4, 2)
    RUN("dot2 2-ilp", k_dot2x2, 2, 4) RUN("dot2 2-ilp", k_dot2x2, 4, 4)
    RUN("Fq2 mul chain", k_fq2mul, 2, 4) RUN("Fq2 mul chain", k_fq2mul, 4, 4)
    {   // mixed-addition chains with gathered operands (table of 2^16 points: L2-resident)
        const uint32_t npts = 1u << 16;
        G2::Affine *pts; hipMalloc(&pts, sizeof(G2::Affine) * npts);
        std::vector<uint32_t> hp(sizeof(G2::Affine) / 4 * npts);
        for (size_t i = 0; i < hp.size(); i++) hp[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
        hipMemcpy(pts, hp.data(), 4 * hp.size(), hipMemcpyHostToDevice);
        void *out; hipMalloc(&out, sizeof(G2::XYZZ) * (size_t)CU * 4 * 4 * 64);
        const int mit = 200;
        { int g = CU * 4 * 4; float ms = time_kernel(k_madd<G1, 4>, g, 64, (const G1::Affine *)pts, (G1::XYZZ *)out, mit, npts);
          printf("%-10s G1 madd chain 4 waves/SIMD: %8.3f ms  %7.3f G madd/s\n", tag, ms, (double)g * 64 * mit / ms * 1e-6); }
        { int g = CU * 4 * 2; float ms = time_kernel(k_madd<G2, 2>, g, 64, (const G2::Affine *)pts, (G2::XYZZ *)out, mit, npts);
          printf("%-10s G2 madd chain 2 waves/SIMD: %8.3f ms  %7.3f G madd/s\n", tag, ms, (double)g * 64 * mit / ms * 1e-6); }
    }
    return 0;
}
