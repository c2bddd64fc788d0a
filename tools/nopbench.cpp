// tools/nopbench.cpp -- times the kernels of a code object built from tools/nopbench_dev.cpp (hipModuleLoad), so that
// two builds of the same source -- e.g. with and without the s_nop hipcc puts behind every inline-asm statement --
// can be compared on one box.  Usage: nopbench a.co [b.co ...]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
static float run(hipFunction_t f, int grid, int block, void **args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, 0, args, 0); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0); hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, 0, args, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main(int argc, char **argv) {
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int CU = p.multiProcessorCount;
    const uint32_t npts = 1u << 16;
    void *io, *pts, *out;
    hipMalloc(&io, 32 * 2 * (size_t)CU * 8 * 256); hipMalloc(&pts, 128 * (size_t)npts); hipMalloc(&out, 256 * (size_t)CU * 16 * 64);
    std::vector<uint32_t> h(16 * (size_t)CU * 8 * 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
    hipMemcpy(io, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    hipMemcpy(pts, h.data(), 128 * (size_t)npts, hipMemcpyHostToDevice);
    for (int a = 1; a < argc; a++) {
        hipModule_t m; if (hipModuleLoad(&m, argv[a]) != hipSuccess) { printf("cannot load %s\n", argv[a]); return 1; }
        hipFunction_t f;
        int it = 400, mit = 200; uint32_t np = npts;
        { void *args[] = {&io, &it};
          hipModuleGetFunction(&f, m, "k_mul_w4"); float ms = run(f, CU * 4, 256, args);
          printf("%-28s mul chain 4 waves/SIMD %8.3f ms %7.2f G mul/s\n", argv[a], ms, 2.0 * CU * 4 * 256 * it / ms * 1e-6);
          hipModuleGetFunction(&f, m, "k_mul_w2"); ms = run(f, CU * 2, 256, args);
          printf("%-28s mul chain 2 waves/SIMD %8.3f ms %7.2f G mul/s\n", argv[a], ms, 2.0 * CU * 2 * 256 * it / ms * 1e-6); }
        { void *args[] = {&pts, &out, &mit, &np};
          hipModuleGetFunction(&f, m, "k_madd_g1"); float ms = run(f, CU * 16, 64, args);
          printf("%-28s G1 madd 4 waves/SIMD   %8.3f ms %7.3f G madd/s\n", argv[a], ms, (double)CU * 16 * 64 * mit / ms * 1e-6);
          hipModuleGetFunction(&f, m, "k_madd_g2"); ms = run(f, CU * 8, 64, args);
          printf("%-28s G2 madd 2 waves/SIMD   %8.3f ms %7.3f G madd/s\n", argv[a], ms, (double)CU * 8 * 64 * mit / ms * 1e-6); }
        hipModuleUnload(m);
    }
    return 0;
}
