// fit_spline.cpp -- the `fitSpline` caller of the LM path (host code, no device code of its own).
//
// Replaces /root/reference/source/mir/optim/fit_splie.d:26-85 (`fitSpline`): fit the VALUES of a cubic spline at
// fixed knots x to scattered points by least squares, optionally with a smoothness penalty, through the
// library's own mir_optimize_least_squares entry (n = knots, m = points (+1): a tiny problem that exercises
// the host-callback path, bounds included).
//
// The spline itself is mir.interpolate.spline (mir-algorithm, an un-vendored dependency, dub.sdl:8):
// `SplineConfiguration!T()` selects the C2 cubic spline with not-a-knot boundaries; it is restated here in
// Hermite form (values + first derivatives), which is also how mir stores it. The reference's own unittest
// (FS:88-141: 10 knots, 10 points, lambda = 0 and 1e-3) pins both the spline and the quirks kept below:
//   * FS:74, FS:77  the penalty integrates `withTwoDerivatives(x)[1]`, i.e. the FIRST derivative, with the
//                   closed form of a piecewise-linear integrand -- not the second derivative the doc names;
//   * FS:62, FS:83  m = points.length + !lambda and y[$-1] is ALWAYS overwritten by the penalty term: with
//                   lambda != 0 the residual of the last point is replaced (that point is ignored).
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/mir_optim_amd.h"

namespace {

// first derivatives of the C2 cubic spline through (x_i, y_i) with not-a-knot ends (n >= 4); n == 3: the
// parabola through the three points; n == 2: the chord; n == 1: 0.
template <typename T>
void c2_not_a_knot(size_t n, const T* x, const T* y, T* d)
{
    if (n == 0) return;
    if (n == 1) { d[0] = 0; return; }
    if (n == 2) { d[0] = d[1] = (y[1] - y[0]) / (x[1] - x[0]); return; }
    if (n == 3) {
        const T h0 = x[1] - x[0], h1 = x[2] - x[1];
        const T s0 = (y[1] - y[0]) / h0, s1 = (y[2] - y[1]) / h1;
        const T c = (s1 - s0) / (h0 + h1);           // second divided difference
        d[0] = s0 - c * h0;
        d[1] = s0 + c * h0;
        d[2] = s1 + c * h1;
        return;
    }
    // tridiagonal system lo[i] d[i-1] + di[i] d[i] + up[i] d[i+1] = r[i]
    std::vector<T> lo(n), di(n), up(n), up2(n, T(0)), r(n);
    auto h = [&](size_t i) { return x[i + 1] - x[i]; };
    auto s = [&](size_t i) { return (y[i + 1] - y[i]) / h(i); };
    for (size_t i = 1; i + 1 < n; ++i) {
        lo[i] = h(i);
        di[i] = 2 * (h(i - 1) + h(i));
        up[i] = h(i - 1);
        r[i] = 3 * (h(i) * s(i - 1) + h(i - 1) * s(i));
    }
    {   // third derivative continuous across x_1
        const T h0 = h(0), h1 = h(1);
        lo[0] = 0; di[0] = h1; up[0] = h0 + h1;
        r[0] = ((3 * h0 + 2 * h1) * h1 * s(0) + h0 * h0 * s(1)) / (h0 + h1);
    }
    {   // ... and across x_{n-2}
        const T ha = h(n - 2), hb = h(n - 3);
        lo[n - 1] = ha + hb; di[n - 1] = hb; up[n - 1] = 0;
        r[n - 1] = (ha * ha * s(n - 3) + (2 * hb + 3 * ha) * hb * s(n - 2)) / (ha + hb);
    }
    // Gaussian elimination with partial pivoting (the not-a-knot rows are not diagonally dominant); row swaps
    // fill one second super-diagonal (up2), as in LAPACK ?gtsv
    for (size_t i = 0; i + 1 < n; ++i) {
        if (std::fabs(lo[i + 1]) > std::fabs(di[i])) {
            std::swap(di[i], lo[i + 1]);
            std::swap(up[i], di[i + 1]);
            std::swap(up2[i], up[i + 1]);
            std::swap(r[i], r[i + 1]);
        }
        const T f = lo[i + 1] / di[i];
        di[i + 1] -= f * up[i];
        up[i + 1] -= f * up2[i];
        r[i + 1] -= f * r[i];
    }
    d[n - 1] = r[n - 1] / di[n - 1];
    d[n - 2] = (r[n - 2] - up[n - 2] * d[n - 1]) / di[n - 2];
    for (size_t k = n - 2; k-- > 0;) d[k] = (r[k] - up[k] * d[k + 1] - up2[k] * d[k + 2]) / di[k];
}

// value, first and second derivative of the Hermite cubic at t; outside [x_0, x_{n-1}] the end pieces extend
template <typename T>
void hermite_eval(size_t n, const T* x, const T* y, const T* d, T t, T out[3])
{
    if (n == 0) { out[0] = out[1] = out[2] = 0; return; }
    if (n == 1) { out[0] = y[0]; out[1] = out[2] = 0; return; }
    size_t lo = 0, hi = n - 1;                       // interval [x_lo, x_lo+1], lo <= n - 2
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (t < x[mid]) hi = mid; else lo = mid;
    }
    const T h = x[lo + 1] - x[lo];
    const T u = (t - x[lo]) / h;
    const T s = (y[lo + 1] - y[lo]) / h;
    // p(u) = y0 + h u (d0 + u (c2 + u c3)) with c2 = 3 s - 2 d0 - d1, c3 = d0 + d1 - 2 s
    const T c2 = 3 * s - 2 * d[lo] - d[lo + 1], c3 = d[lo] + d[lo + 1] - 2 * s;
    out[0] = y[lo] + h * u * (d[lo] + u * (c2 + u * c3));
    out[1] = d[lo] + u * (2 * c2 + 3 * u * c3);
    out[2] = (2 * c2 + 6 * u * c3) / h;
}

template <typename T> struct FitCtx {
    size_t np, nx;
    const T* points;      // np x 2
    const T* x;
    T lambda;
    T (*dist)(T, T);
    std::vector<T> d;     // derivatives of the current spline
};

// FS:60-84: residuals of the spline with values splineY
template <typename T>
void fit_residuals(void* vctx, size_t m, size_t n, const T* splineY, T* y)
{
    auto* c = static_cast<FitCtx<T>*>(vctx);
    (void)n;
    c2_not_a_knot(c->nx, c->x, splineY, c->d.data());                             // FS:64-66
    for (size_t i = 0; i < c->np && i < m; ++i) {                                  // FS:67-68
        T v[3];
        hermite_eval(c->nx, c->x, splineY, c->d.data(), c->points[2 * i], v);
        y[i] = c->dist ? c->dist(v[0], c->points[2 * i + 1]) : v[0] - c->points[2 * i + 1];
    }
    T integral = 0;
    if (c->lambda != 0) {                                                          // FS:71-82
        T ld = c->d[0];                                                            // withTwoDerivatives(x[0])[1]
        for (size_t i = 1; i < c->nx; ++i) {
            const T rd = c->d[i];
            integral += (rd * rd + rd * ld + ld * ld) * (c->x[i] - c->x[i - 1]);
            ld = rd;
        }
    }
    y[m - 1] = std::sqrt(integral * c->lambda * (T)c->np / (T)(3 * c->nx));        // FS:83
}

template <typename T> struct Entry;
template <> struct Entry<double> {
    using S = mir_least_squares_settings_d; using R = mir_least_squares_result_d;
    static R run(const S* s, size_t m, size_t n, double* x, const double* l, const double* u, void* ctx)
    {
        return mir_optimize_least_squares_gpu_d(s, m, n, x, l, u, nullptr, ctx, &fit_residuals<double>, nullptr, nullptr, nullptr, nullptr);
    }
    static void init(S* s) { mir_least_squares_init_d(s); }
};
template <> struct Entry<float> {
    using S = mir_least_squares_settings_s; using R = mir_least_squares_result_s;
    static R run(const S* s, size_t m, size_t n, float* x, const float* l, const float* u, void* ctx)
    {
        return mir_optimize_least_squares_gpu_s(s, m, n, x, l, u, nullptr, ctx, &fit_residuals<float>, nullptr, nullptr, nullptr, nullptr);
    }
    static void init(S* s) { mir_least_squares_init_s(s); }
};

template <typename T>
int fit_spline(const typename Entry<T>::S* settings, size_t np, const T* points, size_t nx, const T* x, const T* l,
               const T* u, T lambda, T (*dist)(T, T), T* splineY, T* splineD, typename Entry<T>::R* result)
{
    if (!points || !x || !l || !u || !splineY || !result || nx == 0 || !(lambda >= 0)) return MIR_FIT_SPLINE_BAD_ARGUMENT;
    if (np < nx && lambda == 0) return MIR_FIT_SPLINE_TOO_FEW_POINTS;              // FS:47-51 (throws there)
    typename Entry<T>::S defaults;
    if (!settings) { Entry<T>::init(&defaults); settings = &defaults; }
    FitCtx<T> c{np, nx, points, x, lambda, dist, std::vector<T>(nx)};
    for (size_t i = 0; i < nx; ++i) splineY[i] = 0;                                // FS:56-57
    const size_t m = np + (lambda == 0 ? 1 : 0);                                   // FS:85
    *result = Entry<T>::run(settings, m, nx, splineY, l, u, &c);
    if (splineD) c2_not_a_knot(nx, x, splineY, splineD);
    return MIR_FIT_SPLINE_OK;
}

}  // namespace

extern "C" {

void mir_spline_c2_derivatives_d(size_t n, const double* x, const double* y, double* d) { c2_not_a_knot(n, x, y, d); }
void mir_spline_c2_derivatives_s(size_t n, const float* x, const float* y, float* d) { c2_not_a_knot(n, x, y, d); }
void mir_spline_eval_d(size_t n, const double* x, const double* y, const double* d, double t, double* out3) { hermite_eval(n, x, y, d, t, out3); }
void mir_spline_eval_s(size_t n, const float* x, const float* y, const float* d, float t, float* out3) { hermite_eval(n, x, y, d, t, out3); }

void mir_fit_spline_residuals_d(size_t npoints, const double* points, size_t nx, const double* x, double lambda,
                                const double* splineY, size_t m, double* y)
{
    FitCtx<double> c{npoints, nx, points, x, lambda, nullptr, std::vector<double>(nx)};
    fit_residuals<double>(&c, m, nx, splineY, y);
}

int mir_fit_spline_d(const mir_least_squares_settings_d* settings, size_t npoints, const double* points, size_t nx,
                     const double* x, const double* l, const double* u, double lambda, double (*dist)(double, double),
                     double* splineY, double* splineD, mir_least_squares_result_d* result)
{
    return fit_spline<double>(settings, npoints, points, nx, x, l, u, lambda, dist, splineY, splineD, result);
}
int mir_fit_spline_s(const mir_least_squares_settings_s* settings, size_t npoints, const float* points, size_t nx,
                     const float* x, const float* l, const float* u, float lambda, float (*dist)(float, float),
                     float* splineY, float* splineD, mir_least_squares_result_s* result)
{
    return fit_spline<float>(settings, npoints, points, nx, x, l, u, lambda, dist, splineY, splineD, result);
}

}  // extern "C"
