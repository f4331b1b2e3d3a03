// A caller's OWN residual model for the batched one-wavefront-per-problem fit (include/mir_optim_amd_batched.hpp): what a
// user of mir_optimize_least_squares_batched_s writes when none of the compiled-in models is theirs. The reference takes an
// arbitrary residual callback (least_squares.d:73-80); here the model is a compile-time type.
// Build (tests/test_gpu_user_model.py does it): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I<repo>/include
#include "mir_optim_amd_batched.hpp"
#include "mir_optim_amd_resident.hpp"

// damped oscillation on a drifting baseline: p0 exp(-p1 t) cos(p2 t + p3) + p4 + p5 sqrt(t)      (n = 6)
// sqrt(t) does not depend on the parameters: it is the row's basis value (tabulated once per launch)
struct DampedCosine {
    static constexpr int n = 6, nb = 1;
    __device__ static void basis(float t, float* b) { b[0] = sqrtf(t); }
    __device__ static float eval(float t, const float* b, const float* x)
    {
        return x[0] * expf(-x[1] * t) * cosf(x[2] * t + x[3]) + x[4] + x[5] * b[0];
    }
    // the reference's optional g callback: d eval / d x_j (used with MIR_LSQ_BATCHED_ANALYTIC_JACOBIAN)
    __device__ static void grad(float t, const float* b, const float* x, float* g)
    {
        const float e = expf(-x[1] * t), ph = x[2] * t + x[3], c = cosf(ph), s = sinf(ph);
        g[0] = e * c;
        g[1] = -t * x[0] * e * c;
        g[2] = -t * x[0] * e * s;
        g[3] = -x[0] * e * s;
        g[4] = 1.0f;
        g[5] = b[0];
    }
};

// every pointer is a DEVICE pointer (the contract of mir_lsq_batched_kernel_s)
extern "C" int user_fit_damped_cosine(const mir_least_squares_settings_s* settings, size_t count, size_t m, float* x,
                                      const float* lower, const float* upper, const float* t, size_t t_stride,
                                      const float* data, mir_least_squares_result_s* results,
                                      const mir_lsq_batched_options* options)
{
    return mir_optim_amd::launch_batched<DampedCosine>(settings, count, m, x, lower, upper, t, t_stride, data, results, options);
}

// ---- the same for the resident-J path (include/mir_optim_amd_resident.hpp: ONE cooperative launch per fit, J in the CUs' LDS):
// logistic growth on a linear baseline, p0 / (1 + exp(-p1 (t - p2))) + p3 + p4 t   (n = 5; per-row data: t and the datum)
struct LogisticGrowth {
    static constexpr int n = 5, nd = 2, nc = 5;
    __device__ static void prepare(const double* x, double* c) { for (int k = 0; k < 5; ++k) c[k] = x[k]; }
    __device__ static double eval(const double* row, const double* c)
    {
        return c[0] / (1.0 + exp(-c[1] * (row[0] - c[2]))) + c[3] + c[4] * row[0] - row[1];
    }
};

// every pointer is a DEVICE pointer; returns launch_resident's code (-3: the slice does not fit a CU's LDS)
extern "C" int user_fit_logistic_resident(const mir_least_squares_settings_d* settings, size_t m, double* x, const double* lower,
                                          const double* upper, const double* rowdata, mir_least_squares_result_d* result,
                                          const mir_lsq_resident_options* options, int* status_out)
{
    return mir_optim_amd::launch_resident<LogisticGrowth>(settings, m, x, lower, upper, rowdata, result, options, status_out);
}
extern "C" size_t user_logistic_workspace_bytes(size_t m) { return mir_optim_amd::resident_workspace_bytes<LogisticGrowth>(m); }
