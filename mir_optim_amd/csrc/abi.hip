// abi.hip -- the C ABI of the library (include/mir_optim_amd.h).
//
// Part 1: the reference's own extern(C) symbols, signature for signature
//   (/root/reference/source/mir/optim/least_squares.d:637-799 "LS", boxcqp.d:36-50 "QP"): work / iwork lengths, status
//   strings, settings init / reset, mir_optimize_least_squares_{d,s}.
// Part 2 (additive): the entry points with mir_lsq_gpu_options, device / stream utilities for bindings without a HIP runtime.
// Everything numeric happens behind solve_entry<T> (solver_loop.hip); this file has no kernels.
#include "driver.h"

using namespace mirlsq;

// ------------------------------------------------------------------------------------------
// ABI pins
// ------------------------------------------------------------------------------------------
static_assert(sizeof(mir_least_squares_settings_d) == 128, "Settings_d must be 128 bytes");
static_assert(offsetof(mir_least_squares_settings_d, jacobianEpsilon) == 8, "");
static_assert(offsetof(mir_least_squares_settings_d, lambdaDecrease) == 96, "");
static_assert(offsetof(mir_least_squares_settings_d, qpSettings) == 104, "");
static_assert(sizeof(mir_least_squares_settings_s) == 68, "Settings_s must be 68 bytes");
static_assert(offsetof(mir_least_squares_settings_s, qpSettings) == 56, "");
static_assert(sizeof(mir_least_squares_result_d) == 32, "Result_d must be 32 bytes");
static_assert(offsetof(mir_least_squares_result_d, residual) == 16, "");
static_assert(sizeof(mir_least_squares_result_s) == 24, "Result_s must be 24 bytes");
static_assert(sizeof(mir_slice_d) == 16 && sizeof(mir_least_squares_task) == 16, "");

extern "C" {

size_t mir_box_qp_work_length(size_t n) { return n * n * 2 + n * 8; }                                  // QP:36-42
size_t mir_box_qp_iwork_length(size_t n) { return n + (n / sizeof(int32_t) + (n % sizeof(int32_t) != 0)); }  // QP:47-50
size_t mir_least_squares_work_length(size_t m, size_t n)                                               // LS:642-646
{
    return mir_box_qp_work_length(n) + n * 5 + n * n + n * m + m * 2;
}
size_t mir_least_squares_iwork_length(size_t m, size_t n)                                              // LS:651-656
{
    (void)m;
    const size_t a = mir_box_qp_iwork_length(n);
    return a > n ? a : n;
}

size_t mir_box_qp_iwork_length_ilp64(size_t n) { return n + (n / sizeof(int64_t) + (n % sizeof(int64_t) != 0)); }   // QP:47-50, lapackint = long
size_t mir_least_squares_iwork_length_ilp64(size_t m, size_t n)
{
    (void)m;
    const size_t a = mir_box_qp_iwork_length_ilp64(n);
    return a > n ? a : n;
}

const char* mir_least_squares_status_string(mir_least_squares_status st)                               // LS:528-557, 666-669
{
    switch (st) {
    case mir_ls_furtherImprovement: return "The algorithm cann't improve the solution";
    case mir_ls_maxIterations: return "Maximum number of iterations reached";
    case mir_ls_xConverged: return "X converged";
    case mir_ls_gConverged: return "Jacobian converged";
    case mir_ls_fConverged: return "Residual is small enough";
    case mir_ls_badBounds: return "Initial guess must be within bounds.";
    case mir_ls_badGuess: return "Initial guess must be an array of finite numbers.";
    case mir_ls_badMinStepQuality: return "0 <= minStepQuality < 1 must hold.";
    case mir_ls_badGoodStepQuality: return "0 < goodStepQuality <= 1 must hold.";
    case mir_ls_badStepQuality: return "minStepQuality < goodStepQuality must hold.";
    case mir_ls_badLambdaParams: return "1 <= lambdaIncrease && lambdaIncrease <= T.max.sqrt and T.min_normal.sqrt <= lambdaDecrease && lambdaDecrease <= 1 must hold.";
    case mir_ls_numericError: return "Numeric Error";
    }
    return "";
}

void mir_least_squares_init_d(mir_least_squares_settings_d* s)                                         // LS:93-122, 761-764
{
    s->maxIterations = 1000; s->maxAge = 0;
    s->jacobianEpsilon = 0x1p-26;            // 2 ^^ ((1 - 53) / 2) (quirk Q10)
    s->absTolerance = DBL_EPSILON; s->relTolerance = 0; s->gradTolerance = DBL_EPSILON;
    s->maxGoodResidual = DBL_EPSILON * DBL_EPSILON;
    s->maxStep = std::sqrt(DBL_MAX) / 16; s->maxLambda = DBL_MAX / 16; s->minLambda = DBL_MIN * 16;
    s->minStepQuality = 0.1; s->goodStepQuality = 0.5; s->lambdaIncrease = 2;
    s->lambdaDecrease = (double)0.30901699437494742410229341718281905886L;   // 1 / (GoldenRatio * 2)
    s->qpSettings.relTolerance = DBL_EPSILON * 16; s->qpSettings.absTolerance = DBL_EPSILON * 16;
    s->qpSettings.maxIterations = 0;
}
void mir_least_squares_init_s(mir_least_squares_settings_s* s)                                         // LS:767-770
{
    s->maxIterations = 1000; s->maxAge = 0;
    s->jacobianEpsilon = 0x1p-11f;           // 2 ^^ ((1 - 24) / 2), integer division
    s->absTolerance = FLT_EPSILON; s->relTolerance = 0; s->gradTolerance = FLT_EPSILON;
    s->maxGoodResidual = FLT_EPSILON * FLT_EPSILON;
    s->maxStep = std::sqrt(FLT_MAX) / 16; s->maxLambda = FLT_MAX / 16; s->minLambda = FLT_MIN * 16;
    s->minStepQuality = 0.1f; s->goodStepQuality = 0.5f; s->lambdaIncrease = 2;
    s->lambdaDecrease = (float)0.30901699437494742410229341718281905886L;
    s->qpSettings.relTolerance = FLT_EPSILON * 16; s->qpSettings.absTolerance = FLT_EPSILON * 16;
    s->qpSettings.maxIterations = 0;
}
void mir_least_squares_reset_d(mir_least_squares_settings_d* s) { mir_least_squares_init_d(s); }        // LS:783-786
void mir_least_squares_reset_s(mir_least_squares_settings_s* s) { mir_least_squares_init_s(s); }        // LS:789-792

mir_least_squares_result_d mir_optimize_least_squares_d(                                               // LS:705-724
    const mir_least_squares_settings_d* settings, size_t m, size_t n, double* x, const double* l, const double* u,
    mir_slice_d work, mir_slice_i iwork, void* fContext, mir_least_squares_function_d f, void* gContext,
    mir_least_squares_jacobian_d g, void* tmContext, mir_least_squares_thread_manager tm)
{
    (void)work; (void)iwork;
    return solve_entry<double>(settings, m, n, x, l, u, nullptr, fContext, f, gContext, g, tmContext, tm);
}

mir_least_squares_result_s mir_optimize_least_squares_s(                                               // LS:729-748
    const mir_least_squares_settings_s* settings, size_t m, size_t n, float* x, const float* l, const float* u,
    mir_slice_s work, mir_slice_i iwork, void* fContext, mir_least_squares_function_s f, void* gContext,
    mir_least_squares_jacobian_s g, void* tmContext, mir_least_squares_thread_manager tm)
{
    (void)work; (void)iwork;
    return solve_entry<float>(settings, m, n, x, l, u, nullptr, fContext, f, gContext, g, tmContext, tm);
}

mir_least_squares_result_d mir_optimize_least_squares_gpu_d(
    const mir_least_squares_settings_d* settings, size_t m, size_t n, double* x, const double* l, const double* u,
    const mir_lsq_gpu_options* options, void* fContext, mir_least_squares_function_d f, void* gContext,
    mir_least_squares_jacobian_d g, void* tmContext, mir_least_squares_thread_manager tm)
{
    return solve_entry<double>(settings, m, n, x, l, u, options, fContext, f, gContext, g, tmContext, tm);
}

mir_least_squares_result_s mir_optimize_least_squares_gpu_s(
    const mir_least_squares_settings_s* settings, size_t m, size_t n, float* x, const float* l, const float* u,
    const mir_lsq_gpu_options* options, void* fContext, mir_least_squares_function_s f, void* gContext,
    mir_least_squares_jacobian_s g, void* tmContext, mir_least_squares_thread_manager tm)
{
    return solve_entry<float>(settings, m, n, x, l, u, options, fContext, f, gContext, g, tmContext, tm);
}

// ---- small device utilities ------------------------------------------------------------------
int mir_lsq_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}
void* mir_lsq_device_malloc(size_t bytes)
{
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
void mir_lsq_device_free(void* p) { if (p) (void)hipFree(p); }
int mir_lsq_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
int mir_lsq_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
int mir_lsq_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream)
{
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -1;
}
void* mir_lsq_stream_create(void)
{
    hipStream_t s = nullptr;
    if (hipStreamCreate(&s) != hipSuccess) return nullptr;
    return s;
}
void mir_lsq_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream)); }
int mir_lsq_stream_synchronize(void* stream) { return hipStreamSynchronize(static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -1; }
const char* mir_lsq_version(void) { return "mir_optim_amd 0.3 (gfx950)"; }

}  // extern "C"
