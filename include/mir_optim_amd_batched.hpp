// mir_optim_amd_batched.hpp -- the batched one-wavefront-per-problem LM fit as a DEVICE HEADER, for residual models of the
// caller's own (HIP C++, gfx950; compile with hipcc -I<repo>/include; the kernel sources under <repo>/mir_optim_amd/csrc travel with it).
//
// The reference takes an arbitrary residual function f (/root/reference/source/mir/optim/least_squares.d:73-80, C tier
// :705-724). The batched kernel (mir_optim_amd/csrc/batched_kernel.h) runs the whole loop of
// optimizeLeastSquaresImplGeneric!T (least_squares.d:877-1176) inside ONE kernel launch with the residual inlined, so a
// function pointer across the FFI is not an option there: the model is a compile-time type instead, and this header is the
// way to hand one in. The three models behind mir_optimize_least_squares_batched_s / mir_lsq_batched_kernel_s
// (MIR_LSQ_MODEL_*) are instances of the same template -- nothing about them is special.
//
// A model:
//     struct MyModel {
//         static constexpr int n  = 4;   // parameters, 1 <= n <= 8
//         static constexpr int nb = 0;   // per-row basis values that do not depend on the parameters (0 = none)
//         __device__ static void  basis(float t, float* b) {}                               // fills b[0 .. nb)
//         __device__ static float eval(float t, const float* b, const float* x)             // model value at t; x[n..8) = 0
//         { return x[0] * __expf(-t * x[1]) * __cosf(x[2] * t) + x[3]; }
//     };
// Optional, the reference's g callback: `__device__ static void grad(float t, const float* b, const float* x, float* g)` -- g[j] =
// d eval / d x_j, j < n -- used instead of finite differences when options->variant has MIR_LSQ_BATCHED_ANALYTIC_JACOBIAN.
// The residual of row i is eval(t_i, basis_i, x) - data_i. eval must be pure (as the reference's callbacks are declared)
// and free of lane-dependent control flow. A problem needs (n + 2) m floats of LDS ((n + 2) m * 4 <= 160 KB - 512).
// Reproducibility: the kernel's own arithmetic is a fixed sequence of IEEE operations (contraction off, every fused multiply-add
// written out: batched_kernel.h), so a fit is reproducible bit for bit on a host (oracle/lm_batched_fused.c does it for the
// built-in cfg 5 model) -- PROVIDED eval is written the same way: `#pragma clang fp contract(off)` as its first statement, explicit
// __builtin_fmaf, and no library transcendental whose bits differ between implementations (mirlsq::det_expf is one that does not).
// An eval written as in the example above is still deterministic on the device; only a host twin would differ in the last bit.
//
//     mir_optim_amd::launch_batched<MyModel>(&settings, count, m, x, lower, upper, t, t_stride, data, results, &options);
// has the contract of mir_lsq_batched_kernel_s (include/mir_optim_amd.h): every pointer a DEVICE pointer, enqueued on
// options->stream, results in place, status -100 (MIR_LSQ_BATCHED_NEEDS_GENERAL) for a problem whose step reaches a finite
// bound. tests/user_model/ holds a complete example that is compiled and compared with the float oracle.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>

#include "mir_optim_amd.h"
#include "../mir_optim_amd/csrc/batched_kernel.h"

namespace mir_optim_amd {

// LDS bytes one problem of `Model` needs at m rows; launch_batched returns -3 when it exceeds kBatchedLdsLimit
constexpr size_t kBatchedLdsLimit = 160 * 1024 - 512;
template <class Model> constexpr size_t batched_lds_bytes(size_t m) { return (size_t)(Model::n + 2) * m * sizeof(float); }
// floats of the per-row basis table a launch needs (0 for a model without a basis): mir_lsq_batched_options.basis
template <class Model> constexpr size_t batched_basis_floats(size_t count, size_t m, size_t t_stride)
{
    return (size_t)Model::nb * (t_stride ? count : 1) * m;
}

// Returns 0, or: -1 bad arguments, -3 a problem does not fit its workgroup's LDS, -4 allocation of the basis table failed,
// -5 the launch failed. Does not synchronise (except in the documented hipMalloc fallback of the basis table).
template <class Model>
int launch_batched(const mir_least_squares_settings_s* S, size_t count, size_t m, float* x, const float* lower, const float* upper,
                   const float* t, size_t t_stride, const float* data, mir_least_squares_result_s* results,
                   const mir_lsq_batched_options* opt = nullptr)
{
    using namespace mirlsq;
    static_assert(Model::n >= 1 && Model::n <= kBatchedNMax, "1 <= n <= 8: one matrix row per lane of a group of eight");
    static_assert(Model::nb >= 0, "nb: number of per-row basis values");
    if (opt && (opt->variant & MIR_LSQ_BATCHED_ANALYTIC_JACOBIAN) && !batched_has_grad<Model>::value) return -1;
    static_assert(sizeof(BatchedResult) == sizeof(mir_least_squares_result_s), "the kernel writes the C result records in place");
    if (!S || !x || !lower || !upper || !t || !data || !results || (t_stride != 0 && t_stride != m)) return -1;
    if (count == 0) return 0;
    const size_t lds = batched_lds_bytes<Model>(m);
    if (m == 0 || lds > kBatchedLdsLimit) return -3;
    hipStream_t stream = opt ? static_cast<hipStream_t>(opt->stream) : nullptr;
    BatchedArgs a{};
    a.set.jacobianEpsilon = S->jacobianEpsilon; a.set.absTolerance = S->absTolerance; a.set.relTolerance = S->relTolerance;
    a.set.gradTolerance = S->gradTolerance; a.set.maxGoodResidual = S->maxGoodResidual; a.set.maxStep = S->maxStep;
    a.set.maxLambda = S->maxLambda; a.set.minLambda = S->minLambda; a.set.minStepQuality = S->minStepQuality;
    a.set.goodStepQuality = S->goodStepQuality; a.set.lambdaIncrease = S->lambdaIncrease; a.set.lambdaDecrease = S->lambdaDecrease;
    a.set.qpRelTolerance = S->qpSettings.relTolerance; a.set.qpAbsTolerance = S->qpSettings.absTolerance;
    a.set.qpMaxIterations = S->qpSettings.maxIterations;
    a.maxIterations = S->maxIterations; a.maxAge = S->maxAge;
    a.count = (int)count; a.m = (int)m; a.t_stride = (int)t_stride;
    a.variant = opt ? opt->variant : 0;
    a.timing = opt ? opt->timing : nullptr;
    a.t = t; a.data = data; a.x = x; a.lower = lower; a.upper = upper;
    a.results = reinterpret_cast<BatchedResult*>(results);
    auto kern = k_lm_batched<Model>;
    if (lds > 48 * 1024
        && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -5;
    float* table = nullptr;
    bool owned = false;
    if constexpr (Model::nb > 0) {
        const size_t rows = (size_t)(t_stride ? count : 1) * m, bytes = rows * Model::nb * sizeof(float);
        if (opt && opt->basis) {
            if (opt->basis_bytes < bytes) return -1;
            table = opt->basis;                                // the caller's table: no allocation in this call
        } else {
            // No table from the caller: hipMalloc, and a stream synchronisation before hipFree below. (Until round 4 this was
            // hipMallocAsync / hipFreeAsync, and 2 of 300 calls with a 2 MB table returned wrong fits for a contiguous range of
            // problems. Root cause, reproduced WITHOUT any library code by scripts/probes/malloc_async_probe.hip on this ROCm
            // (HIP runtime 70226015): with the pool's default release threshold (0) a synchronisation hands the freed block back
            // to the OS, the next hipMallocAsync maps memory at the same address again, and kernels then read wrong words from
            // it -- 84 % of a table per iteration when ordinary hipMalloc / hipFree traffic runs beside it, still some without;
            // with hipMemPoolAttrReleaseThreshold = UINT64_MAX (the pool keeps its memory): none, in any configuration
            // (profiles/r05/malloc_async_probe_*.txt). The runtime's, not this library's; a caller who wants stream-ordered
            // allocation around these launches raises that threshold first. The table here stays in ordinary memory:
            // tests/test_gpu_batched.py::test_repeated_launches_with_a_large_basis_table_agree.)
            owned = true;
            if (hipMalloc((void**)&table, bytes) != hipSuccess) return -4;
        }
        const unsigned bb = (unsigned)std::min<size_t>((rows + 255) / 256, 4096);
        hipLaunchKernelGGL(k_batched_basis<Model>, dim3(bb), dim3(256), 0, stream, t, table, rows);
        a.basis = table;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(64), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (owned) {
        const hipError_t f = hipStreamSynchronize(stream);     // the kernel reads the table: wait before freeing it
        (void)hipFree(table);
        if (e == hipSuccess) e = f;
    }
    return e == hipSuccess ? 0 : -5;
}

// the residual vector of ONE problem, y_i = eval(t_i, basis_i, x) - data_i, as a kernel launch on device pointers: what a
// caller hands to the general solver as its device callback when a batched problem comes back with status -100
template <class Model>
void launch_model_residual(const float* t, const float* data, const float* x, float* y, size_t m, hipStream_t stream)
{
    hipLaunchKernelGGL(mirlsq::k_batched_model_eval<Model>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, t, data, x, y, (int)m);
}

}  // namespace mir_optim_amd
