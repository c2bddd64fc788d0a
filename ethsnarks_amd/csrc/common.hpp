// common.hpp -- runtime include, launch macros and error plumbing for the HIP backend.
#pragma once
#ifdef ZK_EMUL
#include "hip_emul.h"   // tests/emul: CPU stand-in used only by the not-gpu test suite
#define ZK_LAUNCH(kern, grid, block, stream, ...) \
    zk_emul::launch(false, dim3(grid), dim3(block), [&] { kern(__VA_ARGS__); })
#define ZK_LAUNCH_SYNC(kern, grid, block, stream, ...) \
    zk_emul::launch(true, dim3(grid), dim3(block), [&] { kern(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
// every kernel launch goes through this macro: it counts launches, and between zk_profile_begin() / zk_profile_end()
// (a measurement aid of bench.py, off on the proving path) brackets each launch with a HIP event pair on its stream
namespace zk { void launch_pre(const char *name, hipStream_t st); void launch_post(hipStream_t st); }
#ifdef ZK_EXP_MARGINAL
// measurement build only (tools/build_variant.sh marginal -DZK_EXP_MARGINAL): launches whose name contains one of the ';'-separated
// substrings of $ZK_EXP_SKIP are dropped once $ZK_EXP_SKIP_AFTER launches have gone by -- bench.py proves the SAME witness every step, so
// a proof that finds the previous proof's sorted digits / h / bucket sums in its buffers costs what it costs WITHOUT the dropped kernels:
// the marginal cost of a phase inside the full pipeline
namespace zk { bool exp_skip(const char *name); }
#define ZK_LAUNCH(kern, grid, block, stream, ...) \
    do { if (zk::exp_skip(#kern)) break; zk::launch_pre(#kern, stream); hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, __VA_ARGS__); zk::launch_post(stream); } while (0)
#else
#define ZK_LAUNCH(kern, grid, block, stream, ...) \
    do { zk::launch_pre(#kern, stream); hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, __VA_ARGS__); zk::launch_post(stream); } while (0)
#endif
#define ZK_LAUNCH_SYNC ZK_LAUNCH
#endif
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__

#include <stdint.h>
#include <stdio.h>

// error codes of the C ABI (include/zkhip.h)
enum {
    ZK_OK = 0, ZK_ERR_ARG = 1, ZK_ERR_IO = 2, ZK_ERR_FORMAT = 3, ZK_ERR_HIP = 4, ZK_ERR_NOMEM = 5,
    ZK_ERR_SHAPE = 6, ZK_ERR_DEGREE = 7, ZK_ERR_NODEVICE = 8, ZK_ERR_BUFFER = 9, ZK_ERR_INTERNAL = 10
};

namespace zk {
extern thread_local char g_last_error[256];
inline int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    snprintf(g_last_error, sizeof(g_last_error), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return ZK_ERR_HIP;
}
inline int fail_msg(int code, const char *msg) { snprintf(g_last_error, sizeof(g_last_error), "%s", msg); return code; }
}  // namespace zk
// Exception barrier of the C ABI ("returns int codes, never throws / aborts").  Every extern "C" entry point is a
// function-try-block that ends in one of these handlers: the host side uses std::vector / std::string / std::mutex, so an allocation
// failure (a key header that declares 2^28 points, a 2^28-row CSR on a small host) must come back as ZK_ERR_NOMEM, anything else as
// ZK_ERR_INTERNAL with the exception's text in zk_last_error() -- never as std::terminate in the caller's process.
namespace zk { int on_exception() noexcept; }
#define ZK_GUARD catch (...) { return zk::on_exception(); }
#define ZK_GUARD_VOID catch (...) { (void)zk::on_exception(); }
#define ZK_GUARD_BOOL catch (...) { (void)zk::on_exception(); return false; }
#define ZK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return zk::hip_fail(e_, #call, __FILE__, __LINE__); } while (0)
#define ZK_TRY(call) do { int rc_ = (call); if (rc_ != ZK_OK) return rc_; } while (0)

static inline uint32_t zk_div_up(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
