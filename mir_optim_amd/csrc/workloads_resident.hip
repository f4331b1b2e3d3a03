// workloads_resident.hip -- the synthetic workloads of SURVEY.md section 8d as RESIDENT models: the caller's side of
// include/mir_optim_amd_resident.hpp (launch_resident<Model>), compiled into the workloads library like the device callbacks of
// workloads.hip -- the solver library stays free of model code. A model here evaluates the SAME expression as its callback
// kernel there (gauss_row, dtanh), so the two paths minimise the same function; the oracle's copy is oracle/workloads_cpu.c.
//
//   wl_resident_plan(model, m, ...)        how the problem is laid over the chip (or -3: it does not fit)
//   wl_resident_launch_d(model, ...)       launch_resident<Model> on device pointers (asynchronous; what bench.py times)
#include "workloads_device.h"

#include "../../include/mir_optim_amd_resident.hpp"

namespace {

// Gaussian sum: x = [a | c | w | b], n = 3 K + 1; row = {t_i, data_i}; residual sum_k a_k exp(-(t - c_k)^2 / (2 w_k^2)) + b - data.
// Per point: g_k = -1 / (2 w_k w_k) (workloads.hip: gauss_g); per row s = fma(a_k, exp((d d) g_k), s), k ascending from s = b.
template <int K> struct ResGaussSum {
    static constexpr int n = 3 * K + 1, nd = 2, nc = 3 * K + 1;
    __device__ static inline void prepare(const double* x, double* c)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) { c[k] = x[k]; c[K + k] = x[K + k]; c[2 * K + k] = -1.0 / (2 * x[2 * K + k] * x[2 * K + k]); }
        c[3 * K] = x[3 * K];
    }
    __device__ static inline double eval(const double* row, const double* c)
    {
        const double ti = row[0];
        double s = c[3 * K];
#pragma unroll
        for (int k = 0; k < K; ++k) { const double d = ti - c[K + k]; s = fma(c[k], dexp((d * d) * c[2 * K + k]), s); }
        return s - row[1];
    }
};

// tanh-linear: row = {a_i[0 .. N), b_i}; residual tanh(a_i . x) - b_i (cfg 3's family at an LDS-resident size)
template <int N> struct ResTanhLinear {
    static constexpr int n = N, nd = N + 1, nc = N;
    __device__ static inline void prepare(const double* x, double* c)
    {
#pragma unroll
        for (int k = 0; k < N; ++k) c[k] = x[k];
    }
    __device__ static inline double eval(const double* row, const double* c)
    {
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < N; k += 2) { s0 = fma(row[k], c[k], s0); if (k + 1 < N) s1 = fma(row[k + 1], c[k + 1], s1); }
        return dtanh(s0 + s1) - row[N];
    }
    // d/dx_j (tanh(a . x) - b) = (1 - tanh^2(a . x)) a_j
    __device__ static inline void jac(const double* row, const double* c, double* Ji)
    {
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < N; k += 2) { s0 = fma(row[k], c[k], s0); if (k + 1 < N) s1 = fma(row[k + 1], c[k + 1], s1); }
        const double t = dtanh(s0 + s1), d = 1 - t * t;
#pragma unroll
        for (int k = 0; k < N; ++k) Ji[k] = d * row[k];
    }
};

// exponential decay p0 exp(-t / p1) + p2 (reference unittest T5's family, least_squares.d:366-411): a bounded three-parameter fit
struct ResExpDecay1 {
    static constexpr int n = 3, nd = 2, nc = 3;
    __device__ static inline void prepare(const double* x, double* c) { c[0] = x[0]; c[1] = x[1]; c[2] = x[2]; }
    __device__ static inline double eval(const double* row, const double* c) { return c[0] * dexp(-row[0] / c[1]) + c[2] - row[1]; }
};

template <class Model>
int plan_of(size_t m, int num_cu, int* out4, size_t* out2)
{
    mir_optim_amd::ResidentPlan p{};
    const int rc = mir_optim_amd::resident_plan<Model>(m, num_cu, &p);
    if (rc == 0) { out4[0] = p.grid; out4[1] = p.rows; out4[2] = p.groups; out4[3] = Model::n; out2[0] = p.lds_bytes; out2[1] = p.workspace_bytes; }
    return rc;
}

}  // namespace

extern "C" {

enum { WL_RESIDENT_GAUSS5 = 0, WL_RESIDENT_TANH32 = 1, WL_RESIDENT_GAUSS3 = 2, WL_RESIDENT_EXP_DECAY1 = 3 };

// out4 = {grid, rows per workgroup, groups, n}, out2 = {LDS bytes per workgroup, workspace bytes}. Returns 0, -1 (unknown model), -3.
int wl_resident_plan(int model, size_t m, int num_cu, int* out4, size_t* out2)
{
    switch (model) {
    case WL_RESIDENT_GAUSS5: return plan_of<ResGaussSum<5>>(m, num_cu, out4, out2);
    case WL_RESIDENT_TANH32: return plan_of<ResTanhLinear<32>>(m, num_cu, out4, out2);
    case WL_RESIDENT_GAUSS3: return plan_of<ResGaussSum<3>>(m, num_cu, out4, out2);
    case WL_RESIDENT_EXP_DECAY1: return plan_of<ResExpDecay1>(m, num_cu, out4, out2);
    default: return -1;
    }
}

// launch_resident<Model> on DEVICE pointers (include/mir_optim_amd_resident.hpp): x (n, in / out), lower, upper (n), rowdata
// (m x nd), result (one record). Asynchronous on options->stream.
int wl_resident_launch_d(int model, const mir_least_squares_settings_d* settings, size_t m, double* x, const double* lower,
                         const double* upper, const double* rowdata, mir_least_squares_result_d* result,
                         const mir_lsq_resident_options* options, int* status_out)
{
    using namespace mir_optim_amd;
    switch (model) {
    case WL_RESIDENT_GAUSS5: return launch_resident<ResGaussSum<5>>(settings, m, x, lower, upper, rowdata, result, options, status_out);
    case WL_RESIDENT_TANH32: return launch_resident<ResTanhLinear<32>>(settings, m, x, lower, upper, rowdata, result, options, status_out);
    case WL_RESIDENT_GAUSS3: return launch_resident<ResGaussSum<3>>(settings, m, x, lower, upper, rowdata, result, options, status_out);
    case WL_RESIDENT_EXP_DECAY1: return launch_resident<ResExpDecay1>(settings, m, x, lower, upper, rowdata, result, options, status_out);
    default: return -1;
    }
}

}  // extern "C"
