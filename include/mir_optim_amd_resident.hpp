// mir_optim_amd_resident.hpp -- the resident-J Levenberg-Marquardt solver as a DEVICE HEADER (HIP C++, gfx950; compile with
// hipcc -I<repo>/include; the kernel sources under <repo>/mir_optim_amd/csrc travel with it).
//
// For problems whose Jacobian fits the LDS of the chip -- m (n + nd + 3) doubles over the CUs' 160 KB each; BASELINE cfg 2,
// m = 1e5 x n = 16, is 16 MB of 40 -- the whole loop of optimizeLeastSquaresImplGeneric!T
// (/root/reference/source/mir/optim/least_squares.d:877-1176) runs in ONE cooperative kernel launch: every workgroup keeps its
// row slice of J, y and the caller's per-row data in LDS from the first pass to the last (mir_optim_amd/csrc/resident_kernel.h).
// Where the launch-chain path (mir_optimize_least_squares_gpu_d with device callbacks) pays seven kernels of 4-17 us per pass,
// this one pays three in-launch hand-offs. Same algorithm, same statuses and counters, same trace events.
//
// The reference takes an arbitrary residual function f (least_squares.d:73-80). A kernel that keeps the loop on the device
// cannot call back across the FFI, so -- as on the batched path (mir_optim_amd_batched.hpp) -- the model is a compile-time type:
//     struct MyModel {
//         static constexpr int n  = 16;   // parameters, 1 <= n <= 32
//         static constexpr int nd = 2;    // doubles of per-row data (a row of the m x nd row-major table `rowdata`)
//         static constexpr int nc = 16;   // per-POINT constants: what depends on the parameters but not on the row
//         __device__ static void   prepare(const double* x, double* c);          // c[0 .. nc) from x[0 .. n): once per point
//         __device__ static double eval(const double* row, const double* c);     // the RESIDUAL of one row at that point
//     };
// Optional, the reference's g callback (least_squares.d:80): the analytic Jacobian of a row,
//         __device__ static void   jac(const double* row, const double* c, double* Ji);  // Ji[0 .. n) = d residual / d x_j
// used instead of finite differences when options->variant has MIR_LSQ_RESIDENT_ANALYTIC_JACOBIAN (gCalls counts the refreshes,
// the default age limit is 3 as for a caller with a Jacobian, least_squares.d:945).
// prepare / eval must be pure (the reference declares its callbacks pure) and free of thread-dependent control flow;
// `row` and `c` point into LDS. A model with nothing to hoist sets nc = n and copies x.
//
//     mir_optim_amd::launch_resident<MyModel>(&settings, m, x, lower, upper, rowdata, result, &options);
// Every pointer is a DEVICE pointer (x: n, in / out; lower, upper: n; rowdata: m x nd; result: one record); the launch is
// enqueued on options->stream and nothing is synchronised (except in the documented workspace fallback). Returns 0, or:
// -1 bad arguments / settings that fail the reference's validation (the status is then in *status_out), -2 no device,
// -3 the problem does not fit the chip's LDS (use the launch-chain path), -4 allocation failed, -5 the launch failed.
//
// SINGLE TENANT. Every workgroup of the launch must be resident at once (about 89 KB of LDS each at cfg 2: one per CU) and they
// wait for one another with bounded spins (20 s). hipLaunchCooperativeKernel refuses a grid that cannot be resident IN THEORY;
// it cannot know about other work on the device: two processes that each launch a resident solve (or a resident solve beside
// long kernels of another tenant that hold CUs) can each end up half resident until the spins give up. That outcome is a
// SCHEDULING fact, not a numeric one, and it is reported as such: the record's status is numericError (the only failure the
// reference's enum has, LS:132) AND mir_lsq_resident_stats.abort_code != 0 (0 after every solve that ran to its own end,
// including one that ended in a genuine numericError). A caller that may share the device passes `stats`, and on
// abort_code != 0 runs the same problem through the launch chain (mir_optimize_least_squares_gpu_d, ordinary launches that
// wait for nobody) in a fresh launch of the same process -- what mir_optim_amd.workloads.Resident(..., fallback=) does, keeping
// abort_code in the statistics it returns. Same-process solves are safe: ROCm serialises cooperative launches.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>

#include "mir_optim_amd.h"
#include "../mir_optim_amd/csrc/resident_kernel.h"

namespace mir_optim_amd {

struct ResidentPlan {
    int grid;            // workgroups (one per CU at most)
    int rows;            // rows of a workgroup's slice
    int groups;          // group leaders of the reduction
    size_t lds_bytes;    // dynamic LDS of a workgroup
    size_t workspace_bytes;
};

constexpr size_t kResidentLdsLimit = 160 * 1024 - 2048;      // the kernel's static LDS (solve reductions, command) comes on top
constexpr int kResidentMinRows = 64;                          // below this a slice is not worth a workgroup
constexpr size_t kResidentSyncBytes = (2 * mirlsq::kResGroups + 3) * 128;    // counters, flags, seq, abort, look-ahead counter: zeroed before every launch

template <class Model> constexpr int resident_ncb() { return (Model::n + 15) / 16; }

template <class Model> size_t resident_lds_bytes(int rows)
{
    constexpr int NCB = resident_ncb<Model>(), NC = 16 * NCB, NBT = NCB * (NCB + 1) / 2;
    constexpr size_t RPAD = 8 * (mirlsq::res_threads(Model::n) / 64);
    const size_t R = ((size_t)rows + RPAD - 1) / RPAD * RPAD;
    const size_t doubles = R * (NC + 1) + 3 * R + R * Model::nd + 2 * (size_t)Model::n * Model::nc
        + (size_t)(mirlsq::res_threads(Model::n) / 64) * (NBT * 256 + NC) + 3 * NC
        + (Model::n <= 16 ? (size_t)768 : (size_t)mirlsq::LdsSolveCfg<NCB>::ELEMS);      // n <= 16: the one-wave solve's operands and ladder
    return doubles * sizeof(double);
}

namespace detail {
inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct ResidentCarve {
    size_t sync, cmd, look, partial, gtotal, jj0, jj1, jy0, jy1, xs, dx, trial, st, rec, pm, a, fg, vec, ivec, total;
};
template <class Model> ResidentCarve resident_carve(int grid)
{
    using PL = mirlsq::ResPayload<resident_ncb<Model>()>;
    constexpr size_t n = Model::n;
    ResidentCarve c{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += up(bytes, 256); return o; };
    c.sync = take(kResidentSyncBytes);
    c.cmd = take(mirlsq::kResCmdWordsAll * 8);
    c.look = take((size_t)grid * 4 * 8);
    c.partial = take((size_t)grid * PL::STRIDE * 8);
    c.gtotal = take((size_t)mirlsq::kResGroups * PL::STRIDE * 8);
    c.jj0 = take(n * n * 8); c.jj1 = take(n * n * 8);
    c.jy0 = take(n * 8); c.jy1 = take(n * 8);
    c.xs = take(n * 8); c.dx = take(mirlsq::kChainMax * n * 8); c.trial = take(mirlsq::kChainMax * n * 8);
    c.st = take(sizeof(mirlsq::LmState<double>));
    c.rec = take(mirlsq::kChainMax * sizeof(mirlsq::ChainRec<double>));
    c.pm = take(n * n * 8); c.a = take(n * n * 8); c.fg = take(n * (n | 1) * 8);
    c.vec = take(12 * n * 8); c.ivec = take(2 * n * 4);
    c.total = off;
    return c;
}
}  // namespace detail

// How a problem of m rows is laid over `num_cu` CUs. Returns 0, or -3 when a slice does not fit a CU's LDS.
template <class Model> int resident_plan(size_t m, int num_cu, ResidentPlan* plan)
{
    static_assert(Model::n >= 1 && Model::n <= mirlsq::kResNMax, "1 <= n <= 32");
    static_assert(Model::nd >= 0 && Model::nc >= 1 && Model::nc <= mirlsq::kResCMax, "nd: doubles of per-row data, nc: per-point constants (<= 64)");
    if (m == 0 || num_cu < 1) return -1;
    if (num_cu > 256) num_cu = 256;                                   // group leaders sum at most 16 members each
    int grid = (int)((m + kResidentMinRows - 1) / kResidentMinRows);
    if (grid > num_cu) grid = num_cu;
    if (grid < 1) grid = 1;
    const int rows = (int)((m + grid - 1) / grid);
    grid = (int)((m + rows - 1) / rows);                              // no workgroup without rows
    const size_t lds = resident_lds_bytes<Model>(rows);
    if (lds > kResidentLdsLimit) return -3;
    plan->grid = grid;
    plan->rows = rows;
    plan->groups = grid < mirlsq::kResGroups ? grid : mirlsq::kResGroups;
    plan->lds_bytes = lds;
    plan->workspace_bytes = detail::resident_carve<Model>(grid).total;
    return 0;
}
template <class Model> size_t resident_workspace_bytes(size_t m, int num_cu = 256)
{
    ResidentPlan p{};
    return resident_plan<Model>(m, num_cu, &p) == 0 ? p.workspace_bytes : 0;
}

// the reference's validation of the settings, least_squares.d:933-943 (quirk Q9): 0 = fine, else the status to report
inline int resident_check_settings(const mir_least_squares_settings_d* S)
{
    const double dmax = 1.7976931348623157e308, dminn = 2.2250738585072014e-308;
    if (!(0 <= S->minStepQuality && S->minStepQuality < 1)) return mir_ls_badMinStepQuality;
    if (!(0 <= S->goodStepQuality && S->goodStepQuality <= 1)) return mir_ls_badGoodStepQuality;
    if (!(S->minStepQuality < S->goodStepQuality)) return mir_ls_badStepQuality;
    if (!(1 <= S->lambdaIncrease && S->lambdaIncrease <= std::sqrt(dmax))) return mir_ls_badLambdaParams;
    if (!(std::sqrt(dminn) <= S->lambdaDecrease && S->lambdaDecrease <= 1)) return mir_ls_badLambdaParams;
    return 0;
}

template <class Model>
int launch_resident(const mir_least_squares_settings_d* S, size_t m, double* x, const double* lower, const double* upper,
                    const double* rowdata, mir_least_squares_result_d* result, const mir_lsq_resident_options* opt = nullptr,
                    int* status_out = nullptr)
{
    using namespace mirlsq;
    if (!S || !x || !lower || !upper || !result || (Model::nd > 0 && !rowdata) || m == 0 || m > 0x7fffffffu) return -1;
    if (const int bad = resident_check_settings(S)) { if (status_out) *status_out = bad; return -1; }
    if (opt && (opt->variant & MIR_LSQ_RESIDENT_ANALYTIC_JACOBIAN) && !mirlsq::res_has_jac<Model>::value) return -1;
    int dev = 0, num_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess
        || hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu < 1)
        return -2;
    if (opt && opt->max_workgroups && (int)opt->max_workgroups < num_cu) num_cu = (int)opt->max_workgroups;
    ResidentPlan plan{};
    if (const int rc = resident_plan<Model>(m, num_cu, &plan)) return rc;
    hipStream_t stream = opt ? static_cast<hipStream_t>(opt->stream) : nullptr;
    const detail::ResidentCarve c = detail::resident_carve<Model>(plan.grid);
    char* ws = opt ? static_cast<char*>(opt->workspace) : nullptr;
    bool owned = false;
    if (ws) {
        if (opt->workspace_bytes < c.total) return -1;
    } else {
        if (hipMalloc((void**)&ws, c.total) != hipSuccess) return -4;
        owned = true;
    }
    ResidentArgs a{};
    a.set.jacobianEpsilon = S->jacobianEpsilon; a.set.absTolerance = S->absTolerance; a.set.relTolerance = S->relTolerance;
    a.set.gradTolerance = S->gradTolerance; a.set.maxGoodResidual = S->maxGoodResidual; a.set.maxStep = S->maxStep;
    a.set.maxLambda = S->maxLambda; a.set.minLambda = S->minLambda; a.set.minStepQuality = S->minStepQuality;
    a.set.goodStepQuality = S->goodStepQuality; a.set.lambdaIncrease = S->lambdaIncrease; a.set.lambdaDecrease = S->lambdaDecrease;
    a.set.qpRelTolerance = S->qpSettings.relTolerance; a.set.qpAbsTolerance = S->qpSettings.absTolerance;
    a.set.qpMaxIterations = S->qpSettings.maxIterations; a.set.pad = 0;
    a.maxIterations = S->maxIterations; a.maxAge = S->maxAge; a.variant = opt ? opt->variant : 0;
    a.m = (int)m; a.grid = plan.grid; a.rows = plan.rows; a.groups = plan.groups;
    a.rowdata = rowdata; a.x = x; a.lower = lower; a.upper = upper; a.result = result;
    a.cnt = reinterpret_cast<uint32_t*>(ws + c.sync);
    a.flag = a.cnt + 32 * kResGroups;
    a.seq = a.flag + 32 * kResGroups;
    a.abort = a.seq + 32;
    a.lcnt = a.abort + 32;
    a.look = reinterpret_cast<double*>(ws + c.look);
    a.cmd = reinterpret_cast<unsigned long long*>(ws + c.cmd);
    a.partial = reinterpret_cast<double*>(ws + c.partial);
    a.gtotal = reinterpret_cast<double*>(ws + c.gtotal);
    a.JJ[0] = reinterpret_cast<double*>(ws + c.jj0); a.JJ[1] = reinterpret_cast<double*>(ws + c.jj1);
    a.Jy[0] = reinterpret_cast<double*>(ws + c.jy0); a.Jy[1] = reinterpret_cast<double*>(ws + c.jy1);
    a.xs = reinterpret_cast<double*>(ws + c.xs); a.dx = reinterpret_cast<double*>(ws + c.dx);
    a.trial = reinterpret_cast<double*>(ws + c.trial);
    a.st = reinterpret_cast<LmState<double>*>(ws + c.st);
    a.rec = reinterpret_cast<ChainRec<double>*>(ws + c.rec);
    a.sc.Pm = reinterpret_cast<double*>(ws + c.pm); a.sc.A = reinterpret_cast<double*>(ws + c.a);
    a.sc.Fg = reinterpret_cast<double*>(ws + c.fg); a.sc.vec = reinterpret_cast<double*>(ws + c.vec);
    a.sc.ivec = reinterpret_cast<int32_t*>(ws + c.ivec); a.sc.dbg = nullptr;
    // mir_lsq_resident_options is versioned by struct_size: a member is read only when the caller's struct has it
    auto has = [&](size_t offset, size_t size) { return opt && opt->struct_size >= offset + size; };
    a.trace = has(offsetof(mir_lsq_resident_options, trace_records), sizeof(void*)) ? opt->trace_records : nullptr;
    a.trace_capacity = has(offsetof(mir_lsq_resident_options, trace_capacity), sizeof(uint32_t)) ? opt->trace_capacity : 0;
    a.trace_count = has(offsetof(mir_lsq_resident_options, trace_count), sizeof(void*)) ? opt->trace_count : nullptr;
    a.stats = has(offsetof(mir_lsq_resident_options, stats), sizeof(void*)) ? opt->stats : nullptr;

    const bool unbounded = opt && (opt->variant & MIR_LSQ_RESIDENT_UNBOUNDED);
    const void* kern = unbounded ? reinterpret_cast<const void*>(k_lm_resident<Model, false>)
                                 : reinterpret_cast<const void*>(k_lm_resident<Model, true>);
    hipError_t e = hipSuccess;
    if (plan.lds_bytes > 48 * 1024)
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResidentLdsLimit);
    // the polled words start from zero in every launch: a memset node on the stream, ahead of the kernel
    if (e == hipSuccess) e = hipMemsetAsync(ws + c.sync, 0, kResidentSyncBytes, stream);
    if (e == hipSuccess) {
        void* args[] = {&a};
        // cooperative: the launch is refused (hipErrorCooperativeLaunchTooLarge) instead of deadlocking when the grid cannot be
        // resident at once
        e = hipLaunchCooperativeKernel(kern, dim3((unsigned)plan.grid), dim3(res_threads(Model::n)), args, (unsigned)plan.lds_bytes, stream);
    }
    if (owned) {
        const hipError_t f = hipStreamSynchronize(stream);       // the kernel uses the workspace: wait before freeing it
        (void)hipFree(ws);
        if (e == hipSuccess) e = f;
    }
    return e == hipSuccess ? 0 : -5;
}

}  // namespace mir_optim_amd
