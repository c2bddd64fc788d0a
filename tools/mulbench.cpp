// tools/mulbench.cpp -- throughput of the device field multiplication and of the XYZZ mixed addition chains at the
// occupancies the accumulation kernels run at; one binary per arithmetic variant (-DZK_FIPS_V1, -DZK_M_MULLO, ...).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ethsnarks_amd/csrc tools/mulbench.cpp -o tools/mulbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "bn254.hpp"
#include "../../tools/fp52.hpp"
#include <fenv.h>
namespace zk { thread_local char g_last_error[256] = ""; }
using namespace zk;

// ---- FP64 substrate (tools/fp52.hpp): 5 x 52-bit limbs in doubles, hi / lo halves of a limb product from two v_fma_f64 in
// round-toward-zero mode, 64-bit integer column sums.  MODE register, DP rounding field [3:2] <- 3 (toward zero): hwreg(HW_REG_MODE = 1, 2, 2)
// (as inline asm: given the s_setreg BUILTIN the backend's mode-register pass notices the change and puts `s_setreg ... 0` in front of the first
// FP64 instruction of a function that is not strictfp -- the first run of this file computed in round-to-nearest and disagreed with the host)
__device__ __forceinline__ void fp52_rtz() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory"); }
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_f52_mul1(fp52::f52 *io, int iters) {      // one dependent chain per thread
    fp52_rtz();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fp52::f52 a = io[t], b = io[t + gridDim.x * blockDim.x];
    for (int i = 0; i < iters; i++) { a = fp52::mul(a, b); b = fp52::mul(b, a); }
    io[t] = fp52::add_nofold(a, b);
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_f52_mul2(fp52::f52 *io, int iters) {      // two independent chains per thread
    fp52_rtz();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fp52::f52 a = io[t], b = io[t + gridDim.x * blockDim.x], c = fp52::mul(a, a), d = fp52::mul(b, b);
    for (int i = 0; i < iters; i++) { a = fp52::mul(a, b); c = fp52::mul(c, d); b = fp52::mul(b, a); d = fp52::mul(d, c); }
    io[t] = fp52::add_nofold(fp52::mul(a, b), fp52::mul(c, d));
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_f52_dot2(fp52::f52 *io, int iters) {      // two-term dot products with one reduction (what an Fq2 product is made of)
    fp52_rtz();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fp52::f52 a = io[t], b = io[t + gridDim.x * blockDim.x], c = fp52::mul(a, a), d = fp52::mul(b, b);
    for (int i = 0; i < iters; i++) { a = fp52::mul2(a, b, c, d); c = fp52::mul2(c, d, a, b); }
    io[t] = fp52::mul(a, c);
}
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k_f52_fq2mul(fp52::f52 *io, int iters) {    // Fq2 product chain: c0 = a0 b0 + a1 (-b1), c1 = a0 b1 + a1 b0 (the negation left out: timing)
    fp52_rtz();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fp52::f52 x0 = io[t], x1 = io[t + gridDim.x * blockDim.x], y0 = fp52::mul(x0, x0), y1 = fp52::mul(x1, x1);
    for (int i = 0; i < iters; i++) {
        fp52::f52 n0 = fp52::mul2(x0, y0, x1, y1), n1 = fp52::mul2(x0, y1, x1, y0); x0 = n0; x1 = n1;
        n0 = fp52::mul2(y0, x0, y1, x1); n1 = fp52::mul2(y0, x1, y1, x0); y0 = n0; y1 = n1;
    }
    io[t] = fp52::mul2(x0, y0, x1, y1);
}
// correctness on the device: out[3 t] = a b, out[3 t + 1] = a b + c d (u32 x 8 each) for the host to hold against its own evaluation of the same functions
__global__ void k_f52_check(const uint32_t *in, uint32_t *out, uint32_t n) {
    fp52_rtz();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const fp52::f52 a = fp52::from_u32(in + 32 * t), b = fp52::from_u32(in + 32 * t + 8), c = fp52::from_u32(in + 32 * t + 16), d = fp52::from_u32(in + 32 * t + 24);
    fp52::to_u32(fp52::mul(a, b), out + 16 * t); fp52::to_u32(fp52::mul2(a, b, c, d), out + 16 * t + 8);
}

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
    {   // FP64 substrate: the device results against the host's evaluation of the same functions (the host form is held against big integers in
        // the round-4 record, profiles/r04_fp52_mulbench.txt), then the chains
        const uint32_t n = 4096;
        std::vector<uint32_t> in(32 * n), out(16 * n);
        for (size_t i = 0; i < in.size(); i++) in[i] = (uint32_t)((i + 1) * 2654435761u) ^ (uint32_t)(i >> 3) * 40503u;
        for (uint32_t t = 0; t < 4 * n; t++) { in[8 * t + 7] &= 0x3fffffffu; if (t % 29 == 0) for (int i = 0; i < 8; i++) in[8 * t + i] = i == 7 ? 0x3fffffffu : 0xffffffffu; }
        uint32_t *din, *dout; hipMalloc(&din, 4 * in.size()); hipMalloc(&dout, 4 * out.size());
        hipMemcpy(din, in.data(), 4 * in.size(), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_f52_check, dim3(n / 64), dim3(64), 0, 0, (const uint32_t *)din, dout, n);
        hipMemcpy(out.data(), dout, 4 * out.size(), hipMemcpyDeviceToHost);
        fesetround(FE_TOWARDZERO);
        uint32_t bad = 0;
        for (uint32_t t = 0; t < n; t++) {
            const fp52::f52 a = fp52::from_u32(&in[32 * t]), b = fp52::from_u32(&in[32 * t + 8]), c = fp52::from_u32(&in[32 * t + 16]), d = fp52::from_u32(&in[32 * t + 24]);
            uint32_t r[16]; fp52::to_u32(fp52::mul(a, b), r); fp52::to_u32(fp52::mul2(a, b, c, d), r + 8);
            bad += memcmp(r, &out[16 * t], 64) != 0;
        }
        fesetround(FE_TONEAREST);
        printf("%-10s fp52 device == host on %u products and dot products: %s (%u differ)\n", tag, n, bad ? "NO" : "yes", bad);
        fp52::f52 *fio; hipMalloc(&fio, sizeof(fp52::f52) * 2 * (size_t)maxg * block);
        {
            std::vector<fp52::f52> hf(2 * (size_t)maxg * block);
            for (size_t i = 0; i < hf.size(); i++) { uint32_t v[8]; for (int k = 0; k < 8; k++) v[k] = (uint32_t)((8 * i + k) * 2654435761u) & 0x0fffffffu; hf[i] = fp52::from_u32(v); }
            hipMemcpy(fio, hf.data(), sizeof(fp52::f52) * hf.size(), hipMemcpyHostToDevice);
        }
#define RUNF(name, kern, wps, mulsper) { int g = CU * wps; float ms = time_kernel(kern<wps>, g, block, fio, it); \
        printf("%-10s %-14s %d waves/SIMD: %8.3f ms  %7.2f G Fq-mul-equiv/s\n", tag, name, wps, ms, (double)mulsper * g * block * it / ms * 1e-6); }
        RUNF("f52 mul chain", k_f52_mul1, 1, 2) RUNF("f52 mul chain", k_f52_mul1, 2, 2) RUNF("f52 mul chain", k_f52_mul1, 4, 2) RUNF("f52 mul chain", k_f52_mul1, 8, 2)
        RUNF("f52 mul 2-ilp", k_f52_mul2, 1, 4) RUNF("f52 mul 2-ilp", k_f52_mul2, 2, 4) RUNF("f52 mul 2-ilp", k_f52_mul2, 4, 4)
        RUNF("f52 dot2 chain", k_f52_dot2, 2, 2) RUNF("f52 dot2 chain", k_f52_dot2, 4, 2)
        RUNF("f52 Fq2 mul chain", k_f52_fq2mul, 2, 4) RUNF("f52 Fq2 mul chain", k_f52_fq2mul, 4, 4)
    }
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
