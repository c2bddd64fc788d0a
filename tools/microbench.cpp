// tools/microbench.cpp -- gfx950 instruction-rate probes for the 254-bit field arithmetic design
// (the local guides give no integer-multiply rates).  Build: hipcc -O3 --offload-arch=gfx950 -x hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../ethsnarks_amd/csrc/bn254.hpp"
using namespace zk;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k_rate(uint32_t *out, uint32_t seed, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a0 = seed + t, a1 = seed * 3 + t, a2 = seed * 5 + t, a3 = seed * 7 + t;
    uint64_t c0 = t, c1 = t + 1, c2 = t + 2, c3 = t + 3;
    double d0 = t, d1 = t + 1.5, d2 = t + 2.5, d3 = t + 3.5, dm = 1.0000001;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (MODE == 0) {          // v_mad_u64_u32, 4 independent chains
                c0 = (uint64_t)a0 * (uint32_t)c1 + c0; c1 = (uint64_t)a1 * (uint32_t)c2 + c1;
                c2 = (uint64_t)a2 * (uint32_t)c3 + c2; c3 = (uint64_t)a3 * (uint32_t)c0 + c3;
            } else if (MODE == 1) {   // v_mul_lo_u32
                a0 = a0 * a1 + 1; a1 = a1 * a2 + 1; a2 = a2 * a3 + 1; a3 = a3 * a0 + 1;
            } else if (MODE == 2) {   // v_mul_hi_u32
                a0 = __umulhi(a0, a1) + 3; a1 = __umulhi(a1, a2) + 5; a2 = __umulhi(a2, a3) + 7; a3 = __umulhi(a3, a0) + 9;
            } else if (MODE == 3) {   // v_mad_u32_u24
                a0 = __umul24(a0, a1) + a2; a1 = __umul24(a1, a2) + a3; a2 = __umul24(a2, a3) + a0; a3 = __umul24(a3, a0) + a1;
            } else if (MODE == 4) {   // v_fma_f64
                d0 = fma(d0, dm, d1); d1 = fma(d1, dm, d2); d2 = fma(d2, dm, d3); d3 = fma(d3, dm, d0);
            } else if (MODE == 5) {   // v_add_u32 baseline
                a0 = a0 + a1; a1 = a1 + a2; a2 = a2 + a3; a3 = a3 + a0;
            } else if (MODE == 6) {   // 64-bit add (v_add_co + v_addc)
                c0 += c1; c1 += c2; c2 += c3; c3 += c0;
            } else if (MODE == 7) {   // v_add_f64
                d0 = d0 + d1; d1 = d1 + d2; d2 = d2 + d3; d3 = d3 + d0;
            }
        }
    }
    out[t] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(c0 ^ c1 ^ c2 ^ c3) ^ (uint32_t)(d0 + d1 + d2 + d3);
}

template <class F>
__global__ void k_fieldmul(fe *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x];
    for (int i = 0; i < iters; i++) { a = F::mul(a, b); b = F::mul(b, a); }
    io[t] = F::add(a, b);
}
__global__ void k_fieldadd(fe *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x];
    for (int i = 0; i < iters; i++) { a = Fq::add(a, b); b = Fq::sub(b, a); }
    io[t] = Fq::add(a, b);
}
template <class C>
__global__ void k_madd(typename C::Affine *pts, typename C::XYZZ *out, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    typename C::XYZZ acc = C::from_affine(pts[t]);
    typename C::Affine q = pts[t + gridDim.x * blockDim.x];
    for (int i = 0; i < iters; i++) { acc = C::madd(acc, q); q.x = acc.X; }   // q.x perturbation keeps the chain honest (off-curve is fine for timing)
    out[t] = acc;
}

template <class K, class... A>
float time_kernel(K k, int grid, int block, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, CUs %d, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    const int grid = 256 * 8, block = 256, iters = 2000;
    uint32_t *out; CHECK(hipMalloc(&out, 4 * grid * block));
    const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24+add", "v_fma_f64", "v_add_u32", "add_u64(2 instr)", "v_add_f64"};
    float ms[8];
    ms[0] = time_kernel(k_rate<0>, grid, block, out, 12345u, iters); ms[1] = time_kernel(k_rate<1>, grid, block, out, 12345u, iters);
    ms[2] = time_kernel(k_rate<2>, grid, block, out, 12345u, iters); ms[3] = time_kernel(k_rate<3>, grid, block, out, 12345u, iters);
    ms[4] = time_kernel(k_rate<4>, grid, block, out, 12345u, iters); ms[5] = time_kernel(k_rate<5>, grid, block, out, 12345u, iters);
    ms[6] = time_kernel(k_rate<6>, grid, block, out, 12345u, iters);
    ms[7] = time_kernel(k_rate<7>, grid, block, out, 12345u, iters);
    for (int m = 0; m < 8; m++) {
        double ops = (double)grid * block * iters * 64.0;
        printf("%-20s %8.3f ms  %8.2f Tops/s  (%.2f lane-ops/clk/CU at 2.4 GHz)\n", names[m], ms[m], ops / ms[m] * 1e-9,
               ops / (ms[m] * 1e-3) / 2.4e9 / p.multiProcessorCount);
    }
    // field multiplication throughput at several occupancies
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        int g = 256 * blocks_per_cu, it = 500;
        fe *io; CHECK(hipMalloc(&io, 32 * 2 * g * block));
        std::vector<uint32_t> h(16 * g * block);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
        CHECK(hipMemcpy(io, h.data(), 4 * h.size(), hipMemcpyHostToDevice));
        float a = time_kernel(k_fieldmul<Fq>, g, block, io, it);
        float b = time_kernel(k_fieldadd, g, block, io, it);
        printf("Fq::mul  %d blk/CU: %.3f ms -> %.2f Gmul/s ; add/sub: %.3f ms -> %.2f Gop/s\n", blocks_per_cu, a,
               2.0 * g * block * it / a * 1e-6, b, 2.0 * g * block * it / b * 1e-6);
        hipFree(io);
    }
    for (int blocks_per_cu : {1, 2, 4}) {
        int g = 256 * blocks_per_cu, it = 200, blk = 256;
        G1::Affine *p1; G1::XYZZ *o1; CHECK(hipMalloc(&p1, 64 * 2 * g * blk)); CHECK(hipMalloc(&o1, 128 * g * blk));
        std::vector<uint32_t> h(32 * g * blk);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
        CHECK(hipMemcpy(p1, h.data(), 4 * h.size(), hipMemcpyHostToDevice));
        float a = time_kernel(k_madd<G1>, g, blk, p1, o1, it);
        printf("G1 madd  %d blk/CU: %.3f ms -> %.2f Gmadd/s\n", blocks_per_cu, a, (double)g * blk * it / a * 1e-6);
        hipFree(p1); hipFree(o1);
        G2::Affine *p2; G2::XYZZ *o2; CHECK(hipMalloc(&p2, 128 * 2 * g * blk)); CHECK(hipMalloc(&o2, 256 * g * blk));
        std::vector<uint32_t> h2(64 * g * blk);
        for (size_t i = 0; i < h2.size(); i++) h2[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
        CHECK(hipMemcpy(p2, h2.data(), 4 * h2.size(), hipMemcpyHostToDevice));
        float b = time_kernel(k_madd<G2>, g, blk, p2, o2, it);
        printf("G2 madd  %d blk/CU: %.3f ms -> %.2f Gmadd/s\n", blocks_per_cu, b, (double)g * blk * it / b * 1e-6);
        hipFree(p2); hipFree(o2);
    }
    return 0;
}
