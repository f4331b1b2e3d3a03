// A caller's OWN residual model for the batched one-wavefront-per-problem fit (include/mir_optim_amd_batched.hpp): what a
// user of mir_optimize_least_squares_batched_s writes when none of the compiled-in models is theirs. The reference takes an
// arbitrary residual callback (least_squares.d:73-80); here the model is a compile-time type.
// Build (tests/test_gpu_user_model.py does it): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I<repo>/include
#include "mir_optim_amd_batched.hpp"

// damped oscillation on a drifting baseline: p0 exp(-p1 t) cos(p2 t + p3) + p4 + p5 sqrt(t)      (n = 6)
// sqrt(t) does not depend on the parameters: it is the row's basis value (tabulated once per launch)
struct DampedCosine {
    static constexpr int n = 6, nb = 1;
    __device__ static void basis(float t, float* b) { b[0] = sqrtf(t); }
    __device__ static float eval(float t, const float* b, const float* x)
    {
        return x[0] * expf(-x[1] * t) * cosf(x[2] * t + x[3]) + x[4] + x[5] * b[0];
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
