// tools/ntt_bench.cpp -- time of the Fr transforms of the witness map alone (no other work on the device):
// the three launches-per-pass shapes the proving path uses at domain 2^logm (batch 3 inverse + scaling of A, B, C; batch 2 forward of
// A, B; batch 1 inverse of A B formed on load, C subtracted on store: six transforms), against the two bounds that apply: Fr
// multiplications (tools/mulbench.cpp) and HBM bytes.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ethsnarks_amd/csrc tools/ntt_bench.cpp -o tools/ntt_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "ntt.hpp"
namespace zk {
thread_local char g_last_error[256] = "";
void launch_pre(const char *, hipStream_t) {}
void launch_post(hipStream_t) {}
}
using namespace zk;

int main(int argc, char **argv) {
    const uint32_t logm = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 20;
    const uint32_t m = 1u << logm;
    NttTables tab;
    if (ntt_tables_create(tab, logm, nullptr) != ZK_OK) { printf("tables: %s\n", g_last_error); return 1; }
    fe *a, *b;
    hipMalloc(&a, sizeof(fe) * 3 * m); hipMalloc(&b, sizeof(fe) * 3 * m);
    std::vector<fe> h(3 * (size_t)m);
    uint64_t s = 88172645463325252ull;
    for (auto &e : h) for (int l = 0; l < 8; l++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; e.l[l] = (uint32_t)s & (l == 7 ? 0x0fffffffu : ~0u); }
    hipMemcpy(a, h.data(), sizeof(fe) * 3 * m, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    NttFuse alt; alt.post_alt = tab.inv_m_zinv; alt.alt_from = 2;
    NttFuse last; last.in2 = a + m; last.sub = b + 2 * (size_t)m;
    struct Case { const char *name; bool inv; const fe *post; uint32_t batch; NttFuse fuse; int extra_muls; } cases[] = {
        {"inverse x3 + scale", true, tab.inv_then_coset, 3, alt, 1}, {"forward x2", false, nullptr, 2, NttFuse(), 0},
        {"inverse x1, A B on load, - C on store", true, tab.icoset_zinv, 1, last, 2}};
    double total = 0;
    for (auto &c : cases) {
        for (int w = 0; w < 2; w++) ntt_run(tab, a, b, c.inv, nullptr, c.post, nullptr, c.batch, m, c.fuse);
        hipEventRecord(e0);
        for (uint32_t r = 0; r < reps; r++) ntt_run(tab, a, b, c.inv, nullptr, c.post, nullptr, c.batch, m, c.fuse);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        double muls = (double)c.batch * m * (logm / 2.0 + c.extra_muls);
        double bytes = (double)c.batch * m * 32 * 2 * ((logm + NTT_TILE_LOG - 1) / NTT_TILE_LOG);
        printf("%-38s logm %u: %.3f ms   butterfly muls %.1f M -> %.1f G mul/s   min HBM bytes %.0f MB -> %.0f GB/s\n",
               c.name, logm, ms, muls / 1e6, muls / ms / 1e6, bytes / 1e6, bytes / ms / 1e6);
        total += ms;
    }
    printf("witness-map transforms total: %.3f ms\n", total);
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
