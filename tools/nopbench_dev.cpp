// device half of tools/nopbench.cpp: plain-named kernels so the host can load this code object with hipModuleLoad
#include <hip/hip_runtime.h>
#include "bn254.hpp"
using namespace zk;
extern "C" __global__ void __launch_bounds__(256, 4) k_mul_w4(fe *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x];
    for (int i = 0; i < iters; i++) { a = Fq::lmul(a, b); b = Fq::lmul(b, a); }
    io[t] = Fq::ladd(a, b);
}
extern "C" __global__ void __launch_bounds__(256, 2) k_mul_w2(fe *io, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = io[t], b = io[t + gridDim.x * blockDim.x];
    for (int i = 0; i < iters; i++) { a = Fq::lmul(a, b); b = Fq::lmul(b, a); }
    io[t] = Fq::ladd(a, b);
}
extern "C" __global__ void __launch_bounds__(64, 4) k_madd_g1(const G1::Affine *pts, G1::XYZZ *out, int iters, uint32_t npts) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    G1::XYZZ acc = G1::from_affine(pts[t % npts]);
    uint32_t j = t * 7 + 1;
    for (int i = 0; i < iters; i++) { acc = G1::madd(acc, pts[j % npts]); j += 13; }
    out[t] = acc;
}
extern "C" __global__ void __launch_bounds__(64, 2) k_madd_g2(const G2::Affine *pts, G2::XYZZ *out, int iters, uint32_t npts) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    G2::XYZZ acc = G2::from_affine(pts[t % npts]);
    uint32_t j = t * 7 + 1;
    for (int i = 0; i < iters; i++) { acc = G2::madd(acc, pts[j % npts]); j += 13; }
    out[t] = acc;
}
