// keygen_impl.hpp -- kernel + launcher of keygen.hpp; instantiated per curve in msm_g1.cpp / msm_g2.cpp
#pragma once
#include "keygen.hpp"

namespace zk {

template <class C>
__global__ void __launch_bounds__(64, C::WAVES_PER_SIMD / 2)
k_batch_mul_base(typename C::Affine base, const fe *__restrict__ scalars, uint32_t n, typename C::Affine *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe s = Fr::from_mont(scalars[i]);
    typename C::XYZZ acc = C::infinity();
    for (int limb = 7; limb >= 0; limb--) {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) if (k == limb) w = s.l[k];
        for (int b = 31; b >= 0; b--) {
            acc = C::dbl(acc);
            if ((w >> b) & 1) acc = C::madd(acc, base);
        }
    }
    out[i] = C::to_affine(acc);
}

template <class C>
int batch_mul_base(const typename C::Affine &base, const fe *d_scalars_mont, uint32_t n,
                   typename C::Affine *d_out, hipStream_t st) {
    if (!n) return ZK_OK;
    ZK_LAUNCH(k_batch_mul_base<C>, zk_div_up(n, 64), 64, st, base, d_scalars_mont, n, d_out);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
