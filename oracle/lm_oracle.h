/*
 * lm_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference algorithm of libmir/mir-optim's
 * `mir.optim.least_squares` hot path, used ONLY as the checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * mir_optim_amd/ may include, link, dlopen or call anything in oracle/.
 *
 * Reference files restated (paths relative to /root/reference):
 *   LS = source/mir/optim/least_squares.d    QP = source/mir/optim/boxcqp.d
 *     lmo_optimize_{d,s}        <- optimizeLeastSquaresImplGeneric!T  LS:877-1176
 *     lmo_solve_box_qp_{d,s}    <- solveBoxQP!T                       QP:122-379
 *     lmo_apply_bounds_{d,s}    <- applyBounds                        QP:404-410
 *     lmo_settings_init_{d,s}   <- LeastSquaresSettings!T.init        LS:85-123, QP:56-71
 *     lmo_work_length etc.      <- LS:642-656, QP:36-50
 *
 * Third-party arithmetic that is NOT in /root/reference (un-vendored, un-pinned
 * dub dependencies: mir-lapack >=1.2.3, mir-blas, mir-algorithm >=3.7.19, over a
 * system CBLAS/LAPACK chosen by dub configuration, dub.sdl:7-8,26-80):
 *   posvx('E','L')  -> restated here from the published Netlib LAPACK 3.x algorithm
 *                      (dposvx = dpoequ + dlaqsy + dpotrf + dpocon + dpotrs + dporfs),
 *                      see lmo_posvx_{d,s}; checked in tests against the LAPACK that
 *                      ships inside scipy (OpenBLAS 0.3.29).
 *   BLAS-1/2/3      -> plain loops (lm_oracle_impl.inc), Netlib semantics.
 *   Summator!(T, Summation.kbn) (mir.math.sum, QP:284) -> Kahan-Babuska-Neumaier.
 *
 * PARITY PIN: the reference cannot be compiled in this image (it is D; there is no
 * ldc2/dmd/gdc/dub) so the oracle is pinned against the 8 known-answer unittests the
 * reference itself holds for this path (T1-T6 LS:217-434, TQ QP:382-402), see
 * tests/test_oracle_reference_kats.py, plus scipy's dposvx for the LAPACK restatement.
 * No reference test pins any intermediate quantity.
 */
#ifndef LM_ORACLE_H
#define LM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* LeastSquaresStatus, LS:20-46 */
enum {
    LMO_maxIterations = -1,
    LMO_furtherImprovement = 0,
    LMO_xConverged = 1,
    LMO_gConverged = 2,
    LMO_fConverged = 3,
    LMO_badBounds = -32,
    LMO_badGuess = -31,
    LMO_badMinStepQuality = -30,
    LMO_badGoodStepQuality = -29,
    LMO_badStepQuality = -28,
    LMO_badLambdaParams = -27,
    LMO_numericError = -26
};

/* BoxQPStatus, QP:18-26 */
enum { LMO_QP_solved = 0, LMO_QP_numericError = 1, LMO_QP_maxIterations = 2 };

/* BoxQPSettings!T, QP:56-71 */
typedef struct { double relTolerance, absTolerance; uint32_t maxIterations; } lmo_qp_settings_d;
typedef struct { float relTolerance, absTolerance; uint32_t maxIterations; } lmo_qp_settings_s;

/* LeastSquaresSettings!T, LS:85-123 (same field order => same layout as the reference) */
typedef struct {
    uint32_t maxIterations, maxAge;
    double jacobianEpsilon, absTolerance, relTolerance, gradTolerance, maxGoodResidual,
           maxStep, maxLambda, minLambda, minStepQuality, goodStepQuality,
           lambdaIncrease, lambdaDecrease;
    lmo_qp_settings_d qpSettings;
} lmo_settings_d;
typedef struct {
    uint32_t maxIterations, maxAge;
    float jacobianEpsilon, absTolerance, relTolerance, gradTolerance, maxGoodResidual,
          maxStep, maxLambda, minLambda, minStepQuality, goodStepQuality,
          lambdaIncrease, lambdaDecrease;
    lmo_qp_settings_s qpSettings;
} lmo_settings_s;

/* LeastSquaresResult!T, LS:128-143 */
typedef struct { int32_t status; uint32_t iterations, fCalls, gCalls; double residual, lambda; } lmo_result_d;
typedef struct { int32_t status; uint32_t iterations, fCalls, gCalls; float residual, lambda; } lmo_result_s;

/* callbacks, LS:78-80 */
typedef void (*lmo_f_d)(void* ctx, size_t m, size_t n, const double* x, double* y);
typedef void (*lmo_g_d)(void* ctx, size_t m, size_t n, const double* x, double* J);
typedef void (*lmo_f_s)(void* ctx, size_t m, size_t n, const float* x, float* y);
typedef void (*lmo_g_s)(void* ctx, size_t m, size_t n, const float* x, float* J);

/* optional per-pass trace (oracle-only debugging/pinning aid; not in the reference).
 * event: 0 = jacobian refreshed (full), 1 = broyden update, 2 = rejected pass,
 *        3 = accepted pass, 4 = step-size guard (LS:1101) */
typedef void (*lmo_trace_fn)(void* tctx, int event, uint32_t iterations, double lambda,
                             double residual, double trialResidual, double dx_dot);

/* optional sharded-sum hook (oracle-only; used by the CPU gloo tests to restate the
 * multi-GPU row-sharded algorithm of SURVEY.md section 8e): called on every buffer whose
 * value is a sum over rows (JJ lower+Jy packed, residual scalars). NULL = single shard. */
typedef void (*lmo_allreduce_fn)(void* actx, double* buf, size_t count);

typedef struct {
    lmo_trace_fn trace; void* trace_ctx;
    lmo_allreduce_fn allreduce; void* allreduce_ctx;
    int use_openblas;        /* 1: route syrk/gemv/ger/posvx to the OpenBLAS inside scipy (cpu_baseline "port" leg) */
} lmo_options;

size_t lmo_box_qp_work_length(size_t n);            /* QP:36-42 */
size_t lmo_box_qp_iwork_length(size_t n);           /* QP:47-50 */
size_t lmo_work_length(size_t m, size_t n);         /* LS:642-646 */
size_t lmo_iwork_length(size_t m, size_t n);        /* LS:651-656 */
const char* lmo_status_string(int st);              /* LS:528-557 */

void lmo_settings_init_d(lmo_settings_d* s);
void lmo_settings_init_s(lmo_settings_s* s);

lmo_result_d lmo_optimize_d(const lmo_settings_d* settings, size_t m, size_t n,
                            double* x, const double* lower, const double* upper,
                            double* work, int32_t* iwork,
                            void* fctx, lmo_f_d f, void* gctx, lmo_g_d g,
                            const lmo_options* opt);
lmo_result_s lmo_optimize_s(const lmo_settings_s* settings, size_t m, size_t n,
                            float* x, const float* lower, const float* upper,
                            float* work, int32_t* iwork,
                            void* fctx, lmo_f_s f, void* gctx, lmo_g_s g,
                            const lmo_options* opt);

/* P: row-major n x n, lower triangle meaningful (QP:109). Returns BoxQPStatus. */
int lmo_solve_box_qp_d(const lmo_qp_settings_d* settings, size_t n, double* P,
                       const double* q, const double* l, const double* u, double* x,
                       int unconstrainedSolution, double* work, int32_t* iwork,
                       int restoreUpperP, int* qp_iterations);
int lmo_solve_box_qp_s(const lmo_qp_settings_s* settings, size_t n, float* P,
                       const float* q, const float* l, const float* u, float* x,
                       int unconstrainedSolution, float* work, int32_t* iwork,
                       int restoreUpperP, int* qp_iterations);

/* Netlib ?posvx(fact='E', uplo='L', nrhs=1) restated. Column-major A(i,j)=a[i+j*lda],
 * lower triangle referenced; a is overwritten by the equilibrated matrix when equed='Y'.
 * work >= 3n, iwork >= n. ferr is NOT estimated (set to 0): it never influences x or info. */
int lmo_posvx_d(int n, double* a, int lda, double* af, int ldaf, char* equed, double* s,
                double* b, double* x, double* rcond, double* ferr, double* berr,
                double* work, int32_t* iwork);
int lmo_posvx_s(int n, float* a, int lda, float* af, int ldaf, char* equed, float* s,
                float* b, float* x, float* rcond, float* ferr, float* berr,
                float* work, int32_t* iwork);

/* The same ?posvx('E','L') in float with every multiply-add FUSED (fmaf), loop for loop as lmo_posvx_s (no condition estimate):
 * the arithmetic of the device's one-row-per-lane solve (csrc/batched_kernel.h, posvx_rows), which is tested against this bit
 * for bit. a: n x n column-major, lower triangle referenced (not overwritten); n <= 16. Returns info (0, or the order of the
 * leading minor that is not positive); *equilibrated reports ?laqsy's decision. */
int lmo_posvx_fused_s(int n, const float* a, int lda, const float* b, float* x, int* equilibrated);

/* lm_batched_fused.c: ONE float fit of BASELINE cfg 5's padded exponential-decay model (n = 8) with the arithmetic of the device's
 * one-wavefront-per-problem kernel (csrc/batched_kernel.h): fused multiply-adds where the kernel fuses, its per-lane partial sums
 * and wave butterfly, its det_expf, lmo_posvx_fused_s for the damped solve. t: m abscissae, basis: m x 4 (sin 2t, cos 2t, sin 5t,
 * cos 5t AS THE DEVICE TABULATED THEM), data: m, x: 8 in/out. The kernel's results are tested against this bit for bit.
 * Returns 0 (out filled; out->status = LMO_BATCHED_NEEDS_GENERAL when a step hits a finite bound) or -1 (allocation). */
#define LMO_BATCHED_NEEDS_GENERAL (-100)
int lmo_optimize_batched_fused_pad8_s(const lmo_settings_s* S, int m, const float* t, const float* basis, const float* data,
                                      float* x, const float* lower, const float* upper, lmo_result_s* out);

void lmo_apply_bounds_d(size_t n, double* x, const double* l, const double* u);
void lmo_apply_bounds_s(size_t n, float* x, const float* l, const float* u);

/* OpenBLAS (scipy.libs) backend control for the cpu_baseline leg. Returns 0 on success. */
int lmo_openblas_load(const char* path);
int lmo_openblas_set_threads(int nthreads);
void lmo_set_omp_threads(int nthreads);

#ifdef __cplusplus
}
#endif
#endif
