// tests/emul/hip_emul.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal single-process stand-in for the HIP runtime so that the kernel *logic* in
// ethsnarks_amd/csrc (indexing, scans, bucket bookkeeping, barrier placement) can be exercised by
// `pytest -m "not gpu"` in a container that has no GPU.  The same kernel sources are compiled with
// g++ -DZK_EMUL into tests/emul/libzkhip_emul.so; a "launch" runs the blocks one after another, the
// threads of a block either sequentially (kernels without block barriers) or as cooperative fibers that
// yield at __syncthreads() (kernels with barriers).  The product (libzkhip.so) never sees this file, the
// Python package never loads the emulation library, and nothing here is ever timed or shipped.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <ucontext.h>
#include <time.h>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
struct uint2 { unsigned x, y; };

namespace zk_emul {
// Workgroups that use __syncthreads() run as cooperative fibers inside the calling OS thread: a fiber runs until its next
// barrier, then the scheduler resumes the next one; one sweep over all fibers completes the barrier.  (Real threads cost a
// context switch per thread per barrier: ~100x slower.)  On x86-64 the switch is six pushes and pops (swapcontext makes a
// signal-mask system call per switch, which dominated the CPU test time); elsewhere ucontext.
#if defined(__x86_64__)
extern "C" void zk_emul_switch(void **save_sp, void *new_sp);
asm(R"(
    .text
    .weak zk_emul_switch
    .type zk_emul_switch,@function
zk_emul_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size zk_emul_switch, .-zk_emul_switch
)");
struct Fiber { void *sp; bool done; };
inline void *g_sched_sp = nullptr;
#else
struct Fiber { ucontext_t ctx; bool done; };
inline ucontext_t g_sched;
#endif
inline Fiber *g_cur = nullptr;
inline std::function<void()> *g_body = nullptr;
inline std::vector<Fiber> g_fibers;
inline std::vector<char> g_stacks;
#ifndef ZK_EMUL_FIBER_STACK
#define ZK_EMUL_FIBER_STACK (256 * 1024)      // (a sanitizer build inflates the kernels' frames: build it with a larger value)
#endif
constexpr size_t FIBER_STACK = ZK_EMUL_FIBER_STACK;
#if defined(__x86_64__)
inline void yield_to_scheduler() { zk_emul_switch(&g_cur->sp, g_sched_sp); }
inline void trampoline() { (*g_body)(); g_cur->done = true; yield_to_scheduler(); __builtin_trap(); }
inline void fiber_init(Fiber &f, char *stack) {
    uintptr_t top = ((uintptr_t)stack + FIBER_STACK) & ~(uintptr_t)15;
    void **sp = (void **)(top - 64);                      // 6 saved registers, the entry address, a null return address
    for (int i = 0; i < 6; i++) sp[i] = nullptr;
    sp[6] = (void *)&trampoline; sp[7] = nullptr;         // after `ret`: rsp = top - 8, as after a call
    f.sp = sp; f.done = false;
}
inline void fiber_resume(Fiber &f) { zk_emul_switch(&g_sched_sp, f.sp); }
#else
inline void yield_to_scheduler() { swapcontext(&g_cur->ctx, &g_sched); }
inline void trampoline() { (*g_body)(); g_cur->done = true; }
inline void fiber_init(Fiber &f, char *stack) {
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = stack; f.ctx.uc_stack.ss_size = FIBER_STACK; f.ctx.uc_link = &g_sched;
    f.done = false;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
}
inline void fiber_resume(Fiber &f) { swapcontext(&g_sched, &f.ctx); }
#endif

template <class Body>
void launch(bool needs_sync, dim3 grid, dim3 block, Body body) {
    if (!needs_sync) {
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++)
                for (unsigned tx = 0; tx < block.x; tx++) {
                    threadIdx = dim3(tx); blockIdx = dim3(bx, by); blockDim = block; gridDim = grid;
                    body();
                }
        return;
    }
    std::function<void()> fn = body;
    g_body = &fn;
    if (g_fibers.size() < block.x) g_fibers.resize(block.x);
    if (g_stacks.size() < (size_t)block.x * FIBER_STACK) g_stacks.resize((size_t)block.x * FIBER_STACK);
    for (unsigned bxy = 0; bxy < grid.x * grid.y; bxy++) {
        const unsigned bx = bxy % grid.x, by = bxy / grid.x;
        for (unsigned tx = 0; tx < block.x; tx++) fiber_init(g_fibers[tx], g_stacks.data() + (size_t)tx * FIBER_STACK);
        bool any = true;
        while (any) {
            any = false;
            for (unsigned tx = 0; tx < block.x; tx++) {
                Fiber &f = g_fibers[tx];
                if (f.done) continue;
                threadIdx = dim3(tx); blockIdx = dim3(bx, by); blockDim = block; gridDim = grid;
                g_cur = &f;
                fiber_resume(f);
                if (!f.done) any = true;
            }
        }
    }
    g_cur = nullptr; g_body = nullptr;
}
}  // namespace zk_emul

inline void __syncthreads() { if (zk_emul::g_cur) zk_emul::yield_to_scheduler(); }
inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicMax(unsigned *p, unsigned v) {
    unsigned old = *p;
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}

// ---- runtime API subset (synchronous)
typedef int hipError_t;
typedef void *hipStream_t;
typedef struct { double t; } *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
inline const char *hipGetErrorString(hipError_t e) { return e ? "emul error" : "ok"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// device model: ZK_EMUL_DEVICES "devices" (default 1) that share the host's memory; what the emulator keeps per device is what the multi-device
// host logic can get wrong -- the calling thread's CURRENT device, the free memory hipMemGetInfo reports, and a count of allocations made while
// the device was current (tests/test_emul_kernels.py drives the three hooks below through ctypes)
namespace zk_emul {
constexpr int MAX_DEV = 8;
inline int g_devices = [] { const char *e = getenv("ZK_EMUL_DEVICES"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n > MAX_DEV ? MAX_DEV : n; }();
inline thread_local int t_device = 0;
inline size_t g_free[MAX_DEV] = {(size_t)1 << 40, (size_t)1 << 40, (size_t)1 << 40, (size_t)1 << 40, (size_t)1 << 40, (size_t)1 << 40, (size_t)1 << 40, (size_t)1 << 40};
inline std::atomic<uint64_t> g_allocs[MAX_DEV];
}
extern "C" __attribute__((used, visibility("default"))) inline void zk_emul_set_free_mem(int dev, size_t bytes) { if (dev >= 0 && dev < zk_emul::MAX_DEV) zk_emul::g_free[dev] = bytes; }
extern "C" __attribute__((used, visibility("default"))) inline int zk_emul_current_device(void) { return zk_emul::t_device; }
extern "C" __attribute__((used, visibility("default"))) inline uint64_t zk_emul_alloc_count(int dev) { return dev >= 0 && dev < zk_emul::MAX_DEV ? zk_emul::g_allocs[dev].load() : 0; }
inline hipError_t hipGetDeviceCount(int *n) { *n = zk_emul::g_devices; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= zk_emul::g_devices) return hipErrorInvalidValue; zk_emul::t_device = d; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = zk_emul::t_device; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
template <class T> hipError_t hipMalloc(T **p, size_t n) { zk_emul::g_allocs[zk_emul::t_device]++; *p = (T *)calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
template <class T> hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) { *p = (T *)calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = zk_emul::g_free[zk_emul::t_device]; *t = (size_t)1 << 40; return hipSuccess; }
enum { hipHostRegisterDefault = 0 };
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; r++) memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipMemset2DAsync(void *d, size_t pitch, int v, size_t width, size_t height, hipStream_t) {
    for (size_t r = 0; r < height; r++) memset((char *)d + r * pitch, v, width);
    return hipSuccess;
}
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(**e)); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); e->t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
