/*
 * lm_oracle.c -- CPU ORACLE (test infrastructure, NOT the product). See lm_oracle.h.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * PARITY PIN: the reference's own known-answer unittests T1-T6 (LS:217-434) and TQ
 * (QP:382-402) -- the D reference itself is unbuildable in this image (no D compiler).
 */
#include "lm_oracle.h"

#include <dlfcn.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- optional OpenBLAS backend (the class of library the reference links through
 *      mir-blas/mir-lapack); symbols carry scipy's `scipy_` prefix ---- */
typedef void (*ob_dgemv_t)(int order, int trans, int m, int n, double alpha, const double* a, int lda,
                           const double* x, int incx, double beta, double* y, int incy);
typedef void (*ob_dger_t)(int order, int m, int n, double alpha, const double* x, int incx,
                          const double* y, int incy, double* a, int lda);
typedef void (*ob_dsyrk_t)(int order, int uplo, int trans, int n, int k, double alpha, const double* a,
                           int lda, double beta, double* c, int ldc);
typedef void (*ob_dposvx_t)(char* fact, char* uplo, int* n, int* nrhs, double* a, int* lda, double* af,
                            int* ldaf, char* equed, double* s, double* b, int* ldb, double* x, int* ldx,
                            double* rcond, double* ferr, double* berr, double* work, int* iwork, int* info,
                            size_t, size_t, size_t);
typedef void (*ob_set_threads_t)(int);

static struct {
    void* handle;
    ob_dgemv_t dgemv;
    ob_dger_t dger;
    ob_dsyrk_t dsyrk;
    ob_dposvx_t dposvx;
    ob_set_threads_t set_threads;
} g_ob;

int lmo_openblas_load(const char* path)
{
    if (g_ob.handle) return 0;
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    g_ob.dgemv = (ob_dgemv_t)dlsym(h, "scipy_cblas_dgemv");
    g_ob.dger = (ob_dger_t)dlsym(h, "scipy_cblas_dger");
    g_ob.dsyrk = (ob_dsyrk_t)dlsym(h, "scipy_cblas_dsyrk");
    g_ob.dposvx = (ob_dposvx_t)dlsym(h, "scipy_dposvx_");
    g_ob.set_threads = (ob_set_threads_t)dlsym(h, "scipy_openblas_set_num_threads");
    if (!g_ob.dgemv || !g_ob.dger || !g_ob.dsyrk || !g_ob.dposvx) {
        memset(&g_ob, 0, sizeof g_ob);
        dlclose(h);
        return -2;
    }
    g_ob.handle = h;
    return 0;
}

/* threads of the oracle's own OpenMP regions (plain-loop syrk, the workload residuals): bench.py's 1-thread row */
#ifdef _OPENMP
#include <omp.h>
void lmo_set_omp_threads(int nthreads) { if (nthreads > 0) omp_set_num_threads(nthreads); }
#else
void lmo_set_omp_threads(int nthreads) { (void)nthreads; }
#endif

int lmo_openblas_set_threads(int nthreads)
{
    if (!g_ob.handle || !g_ob.set_threads) return -1;
    g_ob.set_threads(nthreads);
    return 0;
}

/* ---- workspace sizes and strings ---- */
size_t lmo_box_qp_work_length(size_t n) { return n * n * 2 + n * 8; }                       /* QP:36-42 */
size_t lmo_box_qp_iwork_length(size_t n) { return n + (n / sizeof(int32_t) + (n % sizeof(int32_t) != 0)); } /* QP:47-50 */
size_t lmo_work_length(size_t m, size_t n)                                                  /* LS:642-646 */
{
    return lmo_box_qp_work_length(n) + n * 5 + n * n + n * m + m * 2;
}
size_t lmo_iwork_length(size_t m, size_t n)                                                 /* LS:651-656 */
{
    (void)m;
    size_t a = lmo_box_qp_iwork_length(n);
    return a > n ? a : n;
}

const char* lmo_status_string(int st)                                                       /* LS:528-557 */
{
    switch (st) {
    case LMO_furtherImprovement: return "The algorithm cann't improve the solution";
    case LMO_maxIterations: return "Maximum number of iterations reached";
    case LMO_xConverged: return "X converged";
    case LMO_gConverged: return "Jacobian converged";
    case LMO_fConverged: return "Residual is small enough";
    case LMO_badBounds: return "Initial guess must be within bounds.";
    case LMO_badGuess: return "Initial guess must be an array of finite numbers.";
    case LMO_badMinStepQuality: return "0 <= minStepQuality < 1 must hold.";
    case LMO_badGoodStepQuality: return "0 < goodStepQuality <= 1 must hold.";
    case LMO_badStepQuality: return "minStepQuality < goodStepQuality must hold.";
    case LMO_badLambdaParams: return "1 <= lambdaIncrease && lambdaIncrease <= T.max.sqrt and T.min_normal.sqrt <= lambdaDecrease && lambdaDecrease <= 1 must hold.";
    case LMO_numericError: return "Numeric Error";
    }
    return "";
}

/* ---- double instantiation ---- */
#define T double
#define NAME(x) x##_d
#define T_EPS DBL_EPSILON
#define T_MAX DBL_MAX
#define T_MIN_NORMAL DBL_MIN
#define T_JACOBIAN_EPS 0x1p-26            /* 2 ^^ ((1 - 53) / 2) */
#define T_SQRT sqrt
#define T_FABS fabs
#define T_FMAX fmax
#define T_FMIN fmin
#define LMO_IS_DOUBLE 1
#include "lm_oracle_impl.inc"
#undef T
#undef NAME
#undef T_EPS
#undef T_MAX
#undef T_MIN_NORMAL
#undef T_JACOBIAN_EPS
#undef T_SQRT
#undef T_FABS
#undef T_FMAX
#undef T_FMIN
#undef LMO_IS_DOUBLE

/* ---- float instantiation (the generic algorithm at T = float with the CORRECT m; the
 *      reference's own float entry passes the literal 2 for m, LS:629, quirk Q7) ---- */
#define T float
#define NAME(x) x##_s
#define T_EPS FLT_EPSILON
#define T_MAX FLT_MAX
#define T_MIN_NORMAL FLT_MIN
#define T_JACOBIAN_EPS 0x1p-11f           /* 2 ^^ ((1 - 24) / 2), integer division */
#define T_SQRT sqrtf
#define T_FABS fabsf
#define T_FMAX fmaxf
#define T_FMIN fminf
#define LMO_IS_DOUBLE 0
#include "lm_oracle_impl.inc"

/* ---- ?posvx('E','L'), float, fused multiply-adds (see lm_oracle.h): the loops of lmo_posvx_s / potrf_lower / potrs_lower with
 *      `s -= a * b` written as fmaf(-a, b, s) and `w += |a| |x|` as fmaf(|a|, |x|, w) ---- */
int lmo_posvx_fused_s(int n, const float* a_in, int lda, const float* b_in, float* x, int* equilibrated)
{
    enum { NMAXF = 16 };
    if (n <= 0 || n > NMAXF) return -1;
    const float eps = FLT_EPSILON / 2, safmin = FLT_MIN;
    float a[NMAXF][NMAXF], f[NMAXF][NMAXF], s[NMAXF], b[NMAXF], r[NMAXF];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) a[i][j] = (j <= i) ? a_in[i + (size_t)j * lda] : a_in[j + (size_t)i * lda];
    float smin = a[0][0], amax = a[0][0];
    for (int i = 0; i < n; ++i) { s[i] = a[i][i]; if (s[i] < smin) smin = s[i]; if (s[i] > amax) amax = s[i]; }
    int rcequ = 0;
    if (smin > 0) {
        const float scond = sqrtf(smin) / sqrtf(amax);
        for (int i = 0; i < n; ++i) s[i] = 1.0f / sqrtf(s[i]);
        const float small = safmin / FLT_EPSILON, large = 1.0f / small;
        rcequ = !(scond >= 0.1f && amax >= small && amax <= large);
    } else {
        for (int i = 0; i < n; ++i) s[i] = 1.0f;
    }
    if (equilibrated) *equilibrated = rcequ;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) { if (rcequ) a[i][j] = s[j] * s[i] * a[i][j]; f[i][j] = a[i][j]; }
        b[i] = rcequ ? s[i] * b_in[i] : b_in[i];
    }
    for (int j = 0; j < n; ++j) {                                    /* ?potf2 'L' */
        float ajj = f[j][j];
        for (int k = 0; k < j; ++k) ajj = fmaf(-f[j][k], f[j][k], ajj);
        if (!(ajj > 0)) return j + 1;
        ajj = sqrtf(ajj);
        f[j][j] = ajj;
        for (int i = j + 1; i < n; ++i) {
            float v = f[i][j];
            for (int k = 0; k < j; ++k) v = fmaf(-f[i][k], f[j][k], v);
            f[i][j] = v / ajj;
        }
    }
#define LMO_POTRS_FUSED(v)                                                                        \
    do {                                                                                          \
        for (int i = 0; i < n; ++i) {                                                             \
            float t = (v)[i];                                                                     \
            for (int k = 0; k < i; ++k) t = fmaf(-f[i][k], (v)[k], t);                            \
            (v)[i] = t / f[i][i];                                                                 \
        }                                                                                         \
        for (int i = n - 1; i >= 0; --i) {                                                        \
            float t = (v)[i];                                                                     \
            for (int k = i + 1; k < n; ++k) t = fmaf(-f[k][i], (v)[k], t);                        \
            (v)[i] = t / f[i][i];                                                                 \
        }                                                                                         \
    } while (0)
    for (int i = 0; i < n; ++i) x[i] = b[i];
    LMO_POTRS_FUSED(x);
    const float safe1 = (float)(n + 1) * safmin, safe2 = safe1 / eps;
    float lstres = 3;
    for (int count = 1;; ++count) {                                  /* ?porfs */
        float berr = 0;
        for (int i = 0; i < n; ++i) {
            float ri = b[i], wi = fabsf(b[i]);
            for (int k = 0; k < n; ++k) { ri = fmaf(-a[i][k], x[k], ri); wi = fmaf(fabsf(a[i][k]), fabsf(x[k]), wi); }
            r[i] = ri;
            const float q = (wi > safe2) ? fabsf(ri) / wi : (fabsf(ri) + safe1) / (wi + safe1);
            if (q > berr) berr = q;
        }
        if (berr > eps && 2 * berr <= lstres && count <= 5) {
            LMO_POTRS_FUSED(r);
            for (int i = 0; i < n; ++i) x[i] += r[i];
            lstres = berr;
            continue;
        }
        break;
    }
#undef LMO_POTRS_FUSED
    if (rcequ) for (int i = 0; i < n; ++i) x[i] = s[i] * x[i];
    return 0;
}
