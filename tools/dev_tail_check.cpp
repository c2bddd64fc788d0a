// dev tool: run the one-lane bucket-reduction kernels on random XYZZ data and compare with the same loops on the host
#include "msm_impl.hpp"
#include <stdio.h>
#include <vector>
namespace zk { thread_local char g_last_error[256] = ""; }
using namespace zk;
typedef G1 C;
static bool eq(const fe &a, const fe &b) { for (int i = 0; i < 8; i++) if (a.l[i] != b.l[i]) return false; return true; }
static bool eqp(const C::XYZZ &a, const C::XYZZ &b) { return eq(a.X, b.X) && eq(a.Y, b.Y) && eq(a.ZZ, b.ZZ) && eq(a.ZZZ, b.ZZZ); }
__global__ void k_canon(C::XYZZ *p, uint32_t n) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = C::canon(p[i]); }
int main() {
    const uint32_t nb = 1024, K = 4, groups = nb / K;
    std::vector<C::XYZZ> bucket(nb);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto rfe = [&]() { fe t; for (int k = 0; k < 8; k += 2) { uint64_t r = rnd(); t.l[k] = (uint32_t)r; t.l[k + 1] = (uint32_t)(r >> 32); } t.l[7] &= 0x0fffffffu; return Fq::to_mont(t); };
    for (auto &p : bucket) { p.X = rfe(); p.Y = rfe(); p.ZZ = rfe(); p.ZZZ = rfe(); }
    for (uint32_t i = 0; i < nb; i += 7) bucket[i] = C::infinity();
    C::XYZZ *d_bucket, *d_pa, *d_pb;
    hipMalloc(&d_bucket, sizeof(C::XYZZ) * nb); hipMalloc(&d_pa, sizeof(C::XYZZ) * (groups + 1)); hipMalloc(&d_pb, sizeof(C::XYZZ) * (groups + 1));
    hipMemcpy(d_bucket, bucket.data(), sizeof(C::XYZZ) * nb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_msm_group_reduce<C, 1>), dim3(zk_div_up(groups, 64)), dim3(64), 0, 0, (const C::XYZZ *)d_bucket, nb, K, d_pa);
    hipLaunchKernelGGL(k_canon, dim3(zk_div_up(groups, 64)), dim3(64), 0, 0, d_pa, groups);
    std::vector<C::XYZZ> part(groups);
    hipMemcpy(part.data(), d_pa, sizeof(C::XYZZ) * groups, hipMemcpyDeviceToHost);
    int bad_g = 0;
    std::vector<C::XYZZ> hpart(groups);
    for (uint32_t g = 0; g < groups; g++) {
        const C::XYZZ *B = bucket.data() + (size_t)g * K;
        C::XYZZ run = C::infinity(), acc = C::infinity();
        for (uint32_t j = K; j-- > 0;) { run = C::add(run, B[j]); acc = C::add(acc, run); }
        if (g) acc = C::add(acc, C::mul_small(run, g * K));
        hpart[g] = acc;
        bad_g += !eqp(part[g], acc);
    }
    // tree sum over the host partials (so a group error does not hide a tree error)
    hipMemcpy(d_pa, hpart.data(), sizeof(C::XYZZ) * groups, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_msm_tree_sum<C, 1>), dim3(1), dim3(MSM_TREE), 0, 0, (const C::XYZZ *)d_pa, groups, d_pb);
    hipLaunchKernelGGL(k_canon, dim3(1), dim3(64), 0, 0, d_pb, 1u);
    C::XYZZ tree;
    hipMemcpy(&tree, d_pb, sizeof(C::XYZZ), hipMemcpyDeviceToHost);
    // host: same tree order (upper half parks, lower half adds)
    std::vector<C::XYZZ> acc(MSM_TREE);
    for (uint32_t i = 0; i < MSM_TREE; i++) acc[i] = i < groups ? hpart[i] : C::infinity();
    for (uint32_t half = MSM_TREE / 2; half > 0; half >>= 1) for (uint32_t t = 0; t < half; t++) acc[t] = C::add(acc[t], acc[t + half]);
    printf("group_reduce mismatches %d of %u; tree_sum %s\n", bad_g, groups, eqp(tree, acc[0]) ? "ok" : "MISMATCH");
    return bad_g || !eqp(tree, acc[0]);
}
