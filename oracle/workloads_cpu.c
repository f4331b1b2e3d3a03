/*
 * workloads_cpu.c -- CPU residual callbacks for the synthetic workloads of SURVEY.md
 * section 8d (ORACLE side: test infrastructure and the cpu_baseline leg only).
 *
 * These are "user callbacks" in the sense of LeastSquaresFunctionBetterC /
 * LeastSquaresJacobianBetterC (/root/reference/source/mir/optim/least_squares.d:78-80):
 *     void f(void* ctx, size_t m, size_t n, const T* x, T* y)
 * Rows are split over OpenMP threads (the reference spreads finite-difference columns over
 * a user thread manager instead, LS:184-215; the arithmetic per element is the same).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* ---- counter-based uniform RNG of SURVEY.md section 8d: u(k) = (splitmix64(seed + k) >> 11) * 2^-53 ---- */
static inline uint64_t splitmix64_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void wlc_uniform(uint64_t seed, uint64_t offset, size_t count, double* out)
{
#pragma omp parallel for schedule(static) if (count > 200000)
    for (ptrdiff_t k = 0; k < (ptrdiff_t)count; ++k)
        out[k] = (double)(splitmix64_mix(seed + offset + (uint64_t)k) >> 11) * 0x1p-53;
}

/* ---- tanh-linear model (cfg 3 / cfg 4): r_i(x) = tanh(a_i . x) - b_i ---- */
typedef struct { const double* A; const double* b; } wlc_tanh_linear_ctx;

void wlc_tanh_linear_f(void* vctx, size_t m, size_t n, const double* x, double* y)
{
    const wlc_tanh_linear_ctx* c = (const wlc_tanh_linear_ctx*)vctx;
#pragma omp parallel for schedule(static) if (m * n > 200000)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)m; ++i) {
        const double* a = c->A + (size_t)i * n;
        double s = 0;
        for (size_t j = 0; j < n; ++j) s += a[j] * x[j];
        y[i] = tanh(s) - c->b[i];
    }
}

void wlc_tanh_linear_g(void* vctx, size_t m, size_t n, const double* x, double* J)
{
    const wlc_tanh_linear_ctx* c = (const wlc_tanh_linear_ctx*)vctx;
#pragma omp parallel for schedule(static) if (m * n > 200000)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)m; ++i) {
        const double* a = c->A + (size_t)i * n;
        double s = 0;
        for (size_t j = 0; j < n; ++j) s += a[j] * x[j];
        double t = tanh(s), d = 1 - t * t;
        double* Ji = J + (size_t)i * n;
        for (size_t j = 0; j < n; ++j) Ji[j] = d * a[j];
    }
}

/* ---- Gaussian-sum curve fit (cfg 2): n = 3K + 1 parameters
 *      x = [a_0..a_{K-1}, c_0..c_{K-1}, w_0..w_{K-1}, b]
 *      r_i = sum_k a_k exp(-(t_i - c_k)^2 / (2 w_k^2)) + b - data_i ---- */
typedef struct { const double* t; const double* data; } wlc_gauss_sum_ctx;

void wlc_gauss_sum_f(void* vctx, size_t m, size_t n, const double* x, double* y)
{
    const wlc_gauss_sum_ctx* c = (const wlc_gauss_sum_ctx*)vctx;
    const size_t K = (n - 1) / 3;
#pragma omp parallel for schedule(static) if (m * n > 200000)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)m; ++i) {
        /* the expression of the device kernels (csrc/workloads.hip: gauss_row): g = -1 / (2 w w), s = fma(a, exp((d d) g), s) */
        double t = c->t[i], s = x[3 * K];
        for (size_t k = 0; k < K; ++k) {
            double d = t - x[K + k], w = x[2 * K + k];
            s = fma(x[k], exp((d * d) * (-1.0 / (2 * w * w))), s);
        }
        y[i] = s - c->data[i];
    }
}

/* ---- exponential decay (reference unittests T4 LS:334-363 and T5 LS:366-411) ----
 *   kind 0: p0 * exp(-t * p1) - data           (T4, n = 2)
 *   kind 1: p0 * exp(-t / p1) + p2 - data      (T5, n = 3) */
typedef struct { const double* t; const double* data; int kind; } wlc_exp_decay_ctx;

void wlc_exp_decay_f(void* vctx, size_t m, size_t n, const double* x, double* y)
{
    const wlc_exp_decay_ctx* c = (const wlc_exp_decay_ctx*)vctx;
    (void)n;
    for (size_t i = 0; i < m; ++i) {
        if (c->kind == 0) y[i] = x[0] * exp(-c->t[i] * x[1]) - c->data[i];
        else y[i] = x[0] * exp(-c->t[i] / x[1]) + x[2] - c->data[i];
    }
}

/* ---- BASELINE cfg 5 family in float: p0 exp(-t p1) + p2 + p3 sin 2t + p4 cos 2t + p5 sin 5t + p6 cos 5t + p7 t - data
 *      (the expression of BatchedModel<kModelExpDecayPad8> in mir_optim_amd/csrc/batched_kernel.h, evaluated with libm) ---- */
typedef struct { const float* t; const float* data; } wlc_curve_ctx_s;

void wlc_exp_pad8_f_s(void* vctx, size_t m, size_t n, const float* x, float* y)
{
    const wlc_curve_ctx_s* c = (const wlc_curve_ctx_s*)vctx;
    (void)n;
    for (size_t i = 0; i < m; ++i) {
        const float t = c->t[i];
        y[i] = x[0] * expf(-t * x[1]) + x[2] + x[3] * sinf(2.0f * t) + x[4] * cosf(2.0f * t) + x[5] * sinf(5.0f * t)
             + x[6] * cosf(5.0f * t) + x[7] * t - c->data[i];
    }
}
