// launch_util.h -- host-side helpers shared by the translation units that launch kernels.
#pragma once

#include <atomic>

#include "common.h"

namespace mirlsq {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: one bit per device ordinal records
// that it has been raised there (devices >= 64 simply set it every time). Safe under concurrent solves.
inline hipError_t ensure_dyn_lds(const void* fn, size_t bytes, std::atomic<uint64_t>& done)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 64;
    const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}
#define MIRLSQ_ENSURE_LDS(kern, bytes)                                                            \
    do {                                                                                          \
        static std::atomic<uint64_t> lds_done_{0};                                                \
        const hipError_t e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), (bytes), lds_done_); \
        if (e_ != hipSuccess) return e_;                                                          \
    } while (0)

}  // namespace mirlsq

// HIP loads a translation unit's device code at the first launch of one of its kernels (tens of ms for the large ones):
// every kernel-carrying translation unit defines preload_<name>() -- one empty kernel -- and mir_lsq_workspace_create calls
// them all, so that the caller's first solve does not pay for it.
#define MIRLSQ_DEFINE_PRELOAD(name)                                                                    \
    namespace { __global__ void k_preload_##name() {} }                                                \
    namespace mirlsq { void preload_##name() { hipLaunchKernelGGL(k_preload_##name, dim3(1), dim3(1), 0, nullptr); } }
namespace mirlsq {
void preload_jtj(); void preload_broyden(); void preload_solve_d(); void preload_solve_s(); void preload_loop(); void preload_jacobian();
void preload_batched();
}
