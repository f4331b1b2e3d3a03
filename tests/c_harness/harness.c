/* harness.c -- a plain C caller of the drop-in boundary, compiled with gcc against include/mir_optim_amd.h and LINKED
 * with libmir_optim_amd.so (no ctypes / libffi in the loop): by-value mir_slice_d / mir_slice_i arguments and the
 * struct result returned through the hidden sret pointer, exactly as a C or C++ user of the reference's
 * mir_optimize_least_squares_d (least_squares.d:705-724) would call it.
 *
 *   harness sizes    print struct sizes / offsets (compared with the ctypes mirror and the reference layout)
 *   harness helpers  defaults, work lengths, status strings (no GPU needed)
 *   harness t2       reference unittest T2 (LS:248-273): Rosenbrock residuals, finite differences through a C thread manager
 *   harness t3b      reference unittest T3b (LS:321-330): bounded Rosenbrock, analytic Jacobian -> BOXCQP active set
 * Output: one `key value` pair per line. Test infrastructure (tests/test_c_harness.py drives it). */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mir_optim_amd.h"

static unsigned f_calls, g_calls, tm_calls, task_calls;

static void rosenbrock_f(void* ctx, size_t m, size_t n, const double* x, double* y)   /* LS:261-265 */
{
    (void)ctx; (void)m; (void)n;
    ++f_calls;
    y[0] = 10 * (x[1] - x[0] * x[0]);
    y[1] = 1 - x[0];
}

static void rosenbrock_g(void* ctx, size_t m, size_t n, const double* x, double* J)   /* LS:295-301, row-major m x n */
{
    (void)ctx; (void)m; (void)n;
    ++g_calls;
    J[0] = -20 * x[0]; J[1] = 10;
    J[2] = -1;         J[3] = 0;
}

/* LeastSquaresThreadManagerBetterC (LS:672-678): call task(taskContext, totalThreads, threadId, i) for i in [0, count) */
static void serial_manager(void* ctx, uint32_t count, mir_least_squares_task taskContext, mir_least_squares_task_function task)
{
    (void)ctx;
    ++tm_calls;
    for (uint32_t i = 0; i < count; ++i) { ++task_calls; task(taskContext, 1, 0, i); }
}

static int run(int bounded)
{
    mir_least_squares_settings_d s;
    mir_least_squares_init_d(&s);
    const size_t m = 2, n = 2;
    double x[2], l[2], u[2];
    if (bounded) { x[0] = x[1] = 150; l[0] = l[1] = 10; u[0] = u[1] = 200; }
    else { x[0] = -1.2; x[1] = 1; l[0] = l[1] = -INFINITY; u[0] = u[1] = INFINITY; }
    mir_slice_d work;
    mir_slice_i iwork;
    work.length = mir_least_squares_work_length(m, n);
    work.ptr = (double*)malloc(work.length * sizeof(double));
    iwork.length = mir_least_squares_iwork_length(m, n);
    iwork.ptr = (int32_t*)calloc(iwork.length, sizeof(int32_t));
    mir_least_squares_result_d r = mir_optimize_least_squares_d(&s, m, n, x, l, u, work, iwork, NULL, rosenbrock_f,
                                                                NULL, bounded ? rosenbrock_g : NULL,
                                                                NULL, bounded ? NULL : serial_manager);
    printf("status %d\niterations %u\nfCalls %u\ngCalls %u\nresidual %.17g\nlambda %.17g\nx0 %.17g\nx1 %.17g\n",
           r.status, r.iterations, r.fCalls, r.gCalls, r.residual, r.lambda, x[0], x[1]);
    printf("status_string %s\n", mir_least_squares_status_string((mir_least_squares_status)r.status));
    printf("host_f_calls %u\nhost_g_calls %u\ntm_calls %u\ntask_calls %u\n", f_calls, g_calls, tm_calls, task_calls);
    free(work.ptr); free(iwork.ptr);
    return 0;
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "sizes";
    if (!strcmp(mode, "sizes")) {
        printf("settings_d %zu\nsettings_s %zu\nresult_d %zu\nresult_s %zu\nslice_d %zu\ntask %zu\n",
               sizeof(mir_least_squares_settings_d), sizeof(mir_least_squares_settings_s), sizeof(mir_least_squares_result_d),
               sizeof(mir_least_squares_result_s), sizeof(mir_slice_d), sizeof(mir_least_squares_task));
        printf("settings_d_qp %zu\nsettings_s_qp %zu\nresult_d_residual %zu\n", offsetof(mir_least_squares_settings_d, qpSettings),
               offsetof(mir_least_squares_settings_s, qpSettings), offsetof(mir_least_squares_result_d, residual));
        printf("gpu_options %zu\nstats %zu\ntrace_record %zu\noptions_variant %zu\noptions_fbRowMajor %zu\n", sizeof(mir_lsq_gpu_options),
               sizeof(mir_lsq_stats), sizeof(mir_lsq_trace_record), offsetof(mir_lsq_gpu_options, variant),
               offsetof(mir_lsq_gpu_options, fbRowMajor));
        return 0;
    }
    if (!strcmp(mode, "helpers")) {
        mir_least_squares_settings_d s;
        memset(&s, 0xff, sizeof s);
        mir_least_squares_init_d(&s);
        printf("maxIterations %u\njacobianEpsilon %.17g\nlambdaDecrease %.17g\nqp_relTolerance %.17g\n", s.maxIterations,
               s.jacobianEpsilon, s.lambdaDecrease, s.qpSettings.relTolerance);
        printf("work_1e6_128 %zu\niwork_1e6_128 %zu\nqp_work_128 %zu\nqp_iwork_128 %zu\n", mir_least_squares_work_length(1000000, 128),
               mir_least_squares_iwork_length(1000000, 128), mir_box_qp_work_length(128), mir_box_qp_iwork_length(128));
        printf("string_numericError %s\nstring_xConverged %s\n", mir_least_squares_status_string(mir_ls_numericError),
               mir_least_squares_status_string(mir_ls_xConverged));
        return 0;
    }
    if (!strcmp(mode, "t2")) return run(0);
    if (!strcmp(mode, "t3b")) return run(1);
    fprintf(stderr, "unknown mode %s\n", mode);
    return 2;
}
