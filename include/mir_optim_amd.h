/*
 * mir_optim_amd.h -- C ABI of the MI355X-native Levenberg-Marquardt solver.
 *
 * Drop-in boundary for ONE path of libmir/mir-optim: `mir.optim.least_squares`.
 * Part 1 declares exactly the `extern(C)` symbols the reference exports for that path,
 * with ABI-identical structs; each declaration cites the reference interface it replaces
 * (LS = source/mir/optim/least_squares.d, QP = source/mir/optim/boxcqp.d under
 * /root/reference).  Part 2 is additive (device-pointer callbacks, batched residuals,
 * multi-GPU communicator, reusable workspace, statistics); nothing in part 2 exists in the
 * reference.
 *
 * All entry points are re-entrant; no global solver state. Errors never throw or abort:
 * every failure is a negative `status` in the returned struct (LS:132).
 * A process without a usable HIP device gets status = mir_ls_numericError from the solve
 * entry points and a diagnostic on stderr -- there is NO CPU fallback in this library.
 */
#ifndef MIR_OPTIM_AMD_H
#define MIR_OPTIM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================================
 * Part 1 -- the reference's extern(C) surface
 * ====================================================================================== */

/* LeastSquaresStatus, LS:20-46 (32-bit enum) */
typedef enum mir_least_squares_status {
    mir_ls_maxIterations = -1,
    mir_ls_furtherImprovement = 0,
    mir_ls_xConverged = 1,
    mir_ls_gConverged = 2,
    mir_ls_fConverged = 3,
    mir_ls_badBounds = -32,
    mir_ls_badGuess = -31,
    mir_ls_badMinStepQuality = -30,
    mir_ls_badGoodStepQuality = -29,
    mir_ls_badStepQuality = -28,
    mir_ls_badLambdaParams = -27,
    mir_ls_numericError = -26
} mir_least_squares_status;

/* BoxQPStatus, QP:18-26 */
typedef enum mir_box_qp_status {
    mir_box_qp_solved = 0,
    mir_box_qp_numericError = 1,
    mir_box_qp_maxIterations = 2
} mir_box_qp_status;

/* BoxQPSettings!T, QP:56-71 */
typedef struct mir_box_qp_settings_d { double relTolerance, absTolerance; uint32_t maxIterations; } mir_box_qp_settings_d;
typedef struct mir_box_qp_settings_s { float relTolerance, absTolerance; uint32_t maxIterations; } mir_box_qp_settings_s;

/* LeastSquaresSettings!double, LS:85-123: 128 bytes, align 8 */
typedef struct mir_least_squares_settings_d {
    uint32_t maxIterations;     /* @0   LS:94  */
    uint32_t maxAge;            /* @4   LS:96  */
    double jacobianEpsilon;     /* @8   LS:98  */
    double absTolerance;        /* @16  LS:100 */
    double relTolerance;        /* @24  LS:102 */
    double gradTolerance;       /* @32  LS:104 */
    double maxGoodResidual;     /* @40  LS:106 */
    double maxStep;             /* @48  LS:108 */
    double maxLambda;           /* @56  LS:110 */
    double minLambda;           /* @64  LS:112 */
    double minStepQuality;      /* @72  LS:114 */
    double goodStepQuality;     /* @80  LS:116 */
    double lambdaIncrease;      /* @88  LS:118 */
    double lambdaDecrease;      /* @96  LS:120 */
    mir_box_qp_settings_d qpSettings;  /* @104 LS:122 */
} mir_least_squares_settings_d;

/* LeastSquaresSettings!float: 68 bytes, align 4 */
typedef struct mir_least_squares_settings_s {
    uint32_t maxIterations, maxAge;
    float jacobianEpsilon, absTolerance, relTolerance, gradTolerance, maxGoodResidual, maxStep,
          maxLambda, minLambda, minStepQuality, goodStepQuality, lambdaIncrease, lambdaDecrease;
    mir_box_qp_settings_s qpSettings;
} mir_least_squares_settings_s;

/* LeastSquaresResult!T, LS:128-143: 32 bytes (double) / 24 bytes (float), returned by hidden pointer */
typedef struct mir_least_squares_result_d {
    int32_t status;             /* LeastSquaresStatus; default numericError (LS:132) */
    uint32_t iterations, fCalls, gCalls;
    double residual;            /* sum of squares, no 1/2, no sqrt (LS:955, LS:1137) */
    double lambda;
} mir_least_squares_result_d;
typedef struct mir_least_squares_result_s {
    int32_t status;
    uint32_t iterations, fCalls, gCalls;
    float residual, lambda;
} mir_least_squares_result_s;

/* mir.ndslice Slice!(T*) 1-D contiguous = { size_t length; T* ptr } (mir-algorithm; LS:713-714) */
typedef struct mir_slice_d { size_t length; double* ptr; } mir_slice_d;
typedef struct mir_slice_s { size_t length; float* ptr; } mir_slice_s;
typedef struct mir_slice_i { size_t length; int32_t* ptr; } mir_slice_i;   /* lapackint = int */

/* LeastSquaresFunctionBetterC / LeastSquaresJacobianBetterC, LS:78-80. J is row-major m x n. */
typedef void (*mir_least_squares_function_d)(void* context, size_t m, size_t n, const double* x, double* y);
typedef void (*mir_least_squares_jacobian_d)(void* context, size_t m, size_t n, const double* x, double* J);
typedef void (*mir_least_squares_function_s)(void* context, size_t m, size_t n, const float* x, float* y);
typedef void (*mir_least_squares_jacobian_s)(void* context, size_t m, size_t n, const float* x, float* J);

/* LeastSquaresTask (an opaque 16-byte D delegate, LS:560-564), LeastSquaresTaskBetterC (LS:567-572)
 * and LeastSquaresThreadManagerBetterC (LS:672-678). The manager must call
 * task(taskContext, totalThreads, threadId, i) once for every i in [0, count). */
typedef struct mir_least_squares_task { void* context; void* function; } mir_least_squares_task;
typedef void (*mir_least_squares_task_function)(mir_least_squares_task task, uint32_t totalThreads,
                                                uint32_t threadId, uint32_t i);
typedef void (*mir_least_squares_thread_manager)(void* context, uint32_t count,
                                                 mir_least_squares_task taskContext,
                                                 mir_least_squares_task_function task);

/* LS:642-646 */
size_t mir_least_squares_work_length(size_t m, size_t n);
/* LS:651-656 */
size_t mir_least_squares_iwork_length(size_t m, size_t n);
/* QP:36-42 */
size_t mir_box_qp_work_length(size_t n);
/* QP:47-50 */
size_t mir_box_qp_iwork_length(size_t n);
/* The same two integer-workspace lengths for builds of the reference with 64-bit `lapackint` (the `*-ilp` dub
 * configurations, dub.sdl:26-80; QP:49 divides by lapackint.sizeof). This library never dereferences iwork, so an ILP64
 * caller only needs lengths that match what its own allocation code expects; every other symbol is layout-identical
 * (Slice!(lapackint*) is {size_t, pointer} for either width). */
size_t mir_box_qp_iwork_length_ilp64(size_t n);
size_t mir_least_squares_iwork_length_ilp64(size_t m, size_t n);
/* LS:666-669 (strings LS:528-557) */
const char* mir_least_squares_status_string(mir_least_squares_status st);
/* LS:761-770 */
void mir_least_squares_init_d(mir_least_squares_settings_d* settings);
void mir_least_squares_init_s(mir_least_squares_settings_s* settings);
/* LS:783-792 */
void mir_least_squares_reset_d(mir_least_squares_settings_d* settings);
void mir_least_squares_reset_s(mir_least_squares_settings_s* settings);

/* LS:705-724. Host-pointer contract of the reference: x/l/u/work/iwork are host memory, f/g/tm
 * are host functions on host memory. The LM arithmetic runs on the GPU; y (and J when g is
 * given) are staged through pinned host buffers. `work`/`iwork` are accepted for ABI
 * compatibility and are not used (device workspace is allocated internally). */
mir_least_squares_result_d mir_optimize_least_squares_d(
    const mir_least_squares_settings_d* settings, size_t m, size_t n,
    double* x, const double* l, const double* u,
    mir_slice_d work, mir_slice_i iwork,
    void* fContext, mir_least_squares_function_d f,
    void* gContext, mir_least_squares_jacobian_d g,
    void* tmContext, mir_least_squares_thread_manager tm);

/* LS:729-748. NOTE: the reference's float entry passes the literal 2 for m (LS:629, a bug);
 * this implementation uses the caller's m. */
mir_least_squares_result_s mir_optimize_least_squares_s(
    const mir_least_squares_settings_s* settings, size_t m, size_t n,
    float* x, const float* l, const float* u,
    mir_slice_s work, mir_slice_i iwork,
    void* fContext, mir_least_squares_function_s f,
    void* gContext, mir_least_squares_jacobian_s g,
    void* tmContext, mir_least_squares_thread_manager tm);

/* ======================================================================================
 * Part 2 -- additive MI355X-native surface (not in the reference)
 * ====================================================================================== */

typedef struct mir_lsq_comm mir_lsq_comm;            /* row-shard communicator */
typedef struct mir_lsq_workspace mir_lsq_workspace;  /* reusable device workspace */

/* Batched residual callback: evaluate p parameter vectors in one sweep over the user's data.
 * X is p x n row-major, Y is p x m row-major (point k's residual vector at Y + k*m); both are
 * DEVICE pointers; work must be enqueued on the options' stream. It is the GPU analogue of the
 * reference's thread manager (LS:184-215): it lets the finite-difference Jacobian evaluate its
 * 2n perturbed points together instead of one at a time. */
typedef void (*mir_lsq_batched_function_d)(void* context, size_t m, size_t n, size_t p, const double* X, double* Y);
typedef void (*mir_lsq_batched_function_s)(void* context, size_t m, size_t n, size_t p, const float* X, float* Y);
enum {
    MIR_LSQ_DEVICE_CALLBACKS = 1u,   /* f/g/fb receive DEVICE pointers and enqueue on `stream` (no host staging) */
    MIR_LSQ_TIME_KERNELS = 2u        /* bracket the hot kernels with HIP events and fill `stats` */
};

/* `variant` bits of mir_lsq_gpu_options. 0 = the product path. The first six select LITERAL RESTATEMENTS of the reference's
 * operation order that the parity tests compare the product kernels with (results agree to rounding, several bit for bit);
 * the rest are diagnostics. Read per call, never latched; unknown bits are ignored. (Bit positions are those of earlier
 * releases; the switches of experiments that were measured and retired -- profiles/r03/ab_*.txt -- are gone.) */
enum {
    MIR_LSQ_VARIANT_BROYDEN_REWRITE = 1u << 0,   /* Broyden passes rewrite J every pass (LS:1003-1006 literally) instead of
                                                    the read-only sweep with pending rank-one terms (broyden_lr.h) */
    MIR_LSQ_VARIANT_FD_SEPARATE_FILL = 1u << 1,  /* ignore fbRowMajor / fbRowMajorDiff: point-major panel + column-fill pass
                                                    (LS:1041-1047 as a kernel of its own) + plain J^T J */
    MIR_LSQ_VARIANT_NO_SPECULATION = 1u << 4,    /* one trial per pass instead of the lambda ladder */
    MIR_LSQ_VARIANT_NO_NULL_SKIP = 1u << 5,      /* evaluate f also for trials equal to x bit for bit */
    MIR_LSQ_VARIANT_SOLVE_BOUNDED = 1u << 6,     /* always the solve kernel with the BOXCQP loop compiled in */
    MIR_LSQ_VARIANT_SOLVE_GENERIC = 1u << 10,    /* the any-n solve kernel (solve_big.h) also for n <= 256 */
    MIR_LSQ_VARIANT_SOLVE_ONE_WORKGROUP = 1u << 11, /* n > 256: the any-n solve on ONE workgroup per damping level, without the
                                                    helper workgroups that share its factorisation (solve_coop.h) */
    MIR_LSQ_VARIANT_FD_HOST_COLUMNS = 1u << 13,  /* host-callback finite differences column by column (per-slot staging vectors,
                                                    a strided column write and a stream synchronisation per task, under a
                                                    lock) instead of through the pinned point-major panel */
    MIR_LSQ_VARIANT_NO_PIPELINE = 1u << 22,      /* no FUSED rounds (by default, with device callbacks and n <= 256: the next pass's
                                                    Broyden sweep runs speculatively behind a round's trial residual and carries
                                                    the trial's sum of squares -- one all-reduce for both -- and ONE kernel decides
                                                    the trial, applies the pass's n x n side and solves the next system): every
                                                    round kernel by kernel, decision first; bit-identical results either way */
    MIR_LSQ_VARIANT_DEBUG_HELPERS_ABSENT = 1u << 12, /* diagnostic: the helper workgroups of the any-n solve are NOT launched although
                                                    its kernel expects them -- the first job times out (5 s), the rescue launch
                                                    solves the pass on one workgroup, mir_lsq_stats.coop_timeouts counts it and the
                                                    solve goes on without helpers: the one-workgroup result, no hang (tests) */
    MIR_LSQ_VARIANT_DEBUG_SOLVE = 1u << 7,       /* diagnostic: print phase stamps of the solve kernel (stderr) */
    MIR_LSQ_VARIANT_HOST_PROFILE = 1u << 8,      /* diagnostic: print host wall time per category of runtime call (stderr) */
    MIR_LSQ_VARIANT_LR_CAP_SHIFT = 16            /* bits 16..20 (a field, not a switch): fold the pending Broyden terms into J
                                                    after this many updates (1..16; 0 = 16) */
};

typedef struct mir_lsq_stats {
    uint64_t passes, accepted, rejected, step_guard_rejects;
    uint64_t jacobian_full, jacobian_broyden;
    uint64_t jtj_launches;           /* launches of the fused Broyden + J^T J + J^T y kernel */
    uint64_t jtj_broyden_launches;   /* of those, with the Broyden update fused in */
    double jtj_ms;                   /* sum of event-timed durations of those launches (ms) */
    double jtj_broyden_ms;
    double solve_ms;                 /* n x n damped solve kernel */
    uint64_t solve_launches;
    double fd_ms;                    /* finite-difference refresh, host wall clock: host-callback mode only (device-callback
                                        refreshes are asynchronous: see fd_callback_ms) */
    double total_ms;                 /* whole call, host wall clock */
    uint64_t qp_active_set_passes;   /* passes in which BOXCQP's active-set loop ran */
    uint64_t broyden_lr_columns;     /* sum over the Broyden sweeps of the pending columns each one read (broyden_lr.h) */
    double jtj_fd_ms;                /* of jtj_ms: launches with the finite-difference fill fused in (fbRowMajor) */
    uint64_t jtj_fd_launches;
    uint64_t elided_evaluations;     /* trial evaluations not made because trial == x bit for bit (null steps at the end of
                                        a noisy solve; the callbacks are `pure`, LS:73-80, so f(trial) is already known);
                                        fCalls counts them like the reference does */
    /* row-shard exchanges of this call (counted whenever a communicator is attached, also with one rank):
     * [0] packed [J^T J lower | J^T y] after a full refresh or a resynchronisation: n(n+1)/2 + n elements each
     * [1] sweep vector of a Broyden pass + the sum of squares of the trial residual it was run behind (fused rounds: the ONE
     *     exchange of the round; one-by-one rounds leave that entry unused): 2n + 35 elements each
     * [2] sums of squares of residual vectors that did not ride on a sweep (entry, re-solve rounds, one-by-one rounds) */
    uint64_t allreduce_calls[3];
    uint64_t allreduce_elems[3];
    uint64_t broyden_flushes;        /* times the pending rank-one terms were folded into J */
    uint64_t jtj_resyncs;            /* of those, followed by a recomputation of J^T J / J^T y from the flushed J */
    /* the CALLER's device callbacks, event-timed on the solver's stream (MIR_LSQ_TIME_KERNELS, device-callback mode):
     * finite-difference evaluations (f / fb / fbRowMajor; points = parameter vectors evaluated) and trial evaluations */
    double fd_callback_ms;
    uint64_t fd_callback_calls, fd_callback_points;
    double trial_callback_ms;
    uint64_t trial_callback_calls, trial_callback_points;
    /* ---- everything below: written only when mir_lsq_gpu_options.stats_size says the caller's struct has it ----
     * kernels the LIBRARY launched (the caller's callbacks and memory copies are not counted), by the kind of round they
     * belong to: [0] rounds that start with a full Jacobian refresh (LS:1008-1050), [1] rounds that start with a Broyden
     * update (LS:999-1007), [2] rounds that re-solve with a larger lambda after a rejection (J unchanged); rounds[k] counts
     * the rounds of each kind (a round = update, solve(s), trial residual(s), decision). library_launches also counts what
     * belongs to no round (entry, flushes). */
    uint64_t library_launches;
    uint64_t round_launches[3];
    uint64_t rounds[3];
    /* host-callback finite differences (the reference ABI, LS:1018-1049): wall time of the refreshes (thread manager
     * included), time spent inside the caller's f summed over the manager's threads, columns refreshed */
    double fd_host_wall_ms, fd_host_f_ms;
    uint64_t fd_host_columns;
    double host_f_ms;                /* host-callback mode: wall time inside the caller's f for the entry and trial evaluations */
    uint64_t host_f_calls;
    uint64_t fused_rounds;           /* rounds whose tail was fused (speculative sweep + trial sum, one exchange, one kernel for
                                        decision + n x n side + next solve); fused_passes: of those, the ones whose trial was
                                        accepted with a Broyden pass next -- the pass run ahead was the reference's next pass */
    uint64_t fused_passes;
    uint64_t coop_timeouts;          /* n > 256: ladder entries whose helper workgroups did not answer within 5 s (a GPU shared with
                                        other work) and that were solved again on one workgroup by the rescue launch -- the result
                                        is the MIR_LSQ_VARIANT_SOLVE_ONE_WORKGROUP one (2e-8 from the helpers'), the status is not
                                        touched; after the first one a solve stops asking for helpers */
} mir_lsq_stats;
/* Versioning of mir_lsq_stats: the library writes min(stats_size, sizeof(mir_lsq_stats)) bytes. A caller whose options
 * struct has no stats_size member (struct_size < 96), or leaves it 0, gets the layout of its era: 120 bytes (through
 * qp_active_set_passes) below struct_size 80, 144 (through jtj_fd_launches) below 88, 264 (through trial_callback_points)
 * from 88 on. Changelog: fd_ms is 0 in device-callback mode since the fd_callback_* fields exist (the refresh is only
 * enqueued there); *_callback_calls / *_callback_points are always counted, the *_ms next to them need MIR_LSQ_TIME_KERNELS. */

/* Optional per-pass trace (not in the reference; a parity-pinning aid: tests compare it event by event with the
 * oracle's trace). One record per Jacobian update and per executed loop pass, in the reference's order -- passes
 * evaluated speculatively and then discarded are not recorded.
 *   event 0: Jacobian refreshed in full (LS:1008-1050)   1: Broyden update (LS:999-1007)
 *         2: pass rejected (LS:1125-1130)                3: pass accepted (LS:1132-1139)
 *         4: step-size guard (LS:1101-1106)
 * lambda is the damping the pass solved with; residual the current (event 3: the new) sum of squares;
 * trial_residual ||f(trial)||^2 (0 for events 0, 1, 4); dx_dot = dx.dx of the pass (events 0, 1: of the last
 * accepted step). `count` counts every event even when it exceeds `capacity` (only the first `capacity` are stored). */
typedef struct mir_lsq_trace_record {
    int32_t event;
    uint32_t iterations;
    double lambda, residual, trial_residual, dx_dot;
} mir_lsq_trace_record;
typedef struct mir_lsq_trace {
    mir_lsq_trace_record* records;
    uint64_t capacity;
    uint64_t count;
} mir_lsq_trace;

typedef struct mir_lsq_gpu_options {
    uint32_t struct_size;            /* = sizeof(mir_lsq_gpu_options) */
    uint32_t flags;
    void* stream;                    /* hipStream_t; NULL = a stream owned by the call */
    mir_lsq_comm* comm;              /* NULL = single GPU; else m is this rank's row count */
    mir_lsq_workspace* workspace;    /* NULL = allocate/free per call */
    void* fbContext;
    void* fb;                        /* mir_lsq_batched_function_{d,s} or NULL */
    uint32_t fd_batch;               /* max points per fb call (0 = 2n) */
    uint32_t variant;                /* MIR_LSQ_VARIANT_* bits; 0 = product path */
    mir_lsq_stats* stats;            /* optional out */
    mir_lsq_trace* trace;            /* optional out; read only when struct_size covers it (costs one extra
                                        device-to-host copy per pass) */
    void* fbRowMajor;                /* optional mir_lsq_batched_function_{d,s} that writes Y as m x p ROW-major
                                        (point k's residual of row i at Y[i * p + k]), context fbContext. With it the
                                        finite-difference refresh of f64 problems with n <= 256 needs
                                        no separate column-fill pass: the 2n points are evaluated in one call and the
                                        J^T J kernel forms the Jacobian rows from the (+h, -h) pairs while it writes J.
                                        Read only when struct_size covers it; `fb` is still used for lambda-ladder trials */
    void* fbRowMajorDiff;            /* optional mir_lsq_batched_function_d (f64; n <= 128, n = 192, n = 256), context fbContext: called with the
                                        p = 2n finite-difference points X = [x + h e_0, x - h e_0, x + h e_1, ...] and writes the
                                        m x n ROW-major DIFFERENCE panel D[i * n + j] = f(X_2j)_i - f(X_2j+1)_i -- the caller's
                                        kernel does the reference's copy + axpy(-1) (LS:1041, 1045) on its way out, every one of
                                        the 2n residual vectors is still evaluated. The panel between the caller's kernel and
                                        the library's is then m x n instead of m x 2n: 2 GB less HBM traffic per refresh at
                                        m = 1e6, n = 128 (1 GB not written, 1 GB not read), bitwise the same Jacobian.
                                        Preferred over fbRowMajor when both are given. Read only when struct_size covers it */
    uint32_t stats_size;             /* sizeof(mir_lsq_stats) as the CALLER compiled it: the library never writes past it
                                        (0 or not covered by struct_size: see "Versioning of mir_lsq_stats") */
    uint32_t reserved0;              /* must be 0 */
} mir_lsq_gpu_options;

/* Same algorithm and result contract as mir_optimize_least_squares_{d,s}; x/l/u stay host
 * pointers (n is small); callbacks follow `options->flags`. `options` may be NULL. */
mir_least_squares_result_d mir_optimize_least_squares_gpu_d(
    const mir_least_squares_settings_d* settings, size_t m, size_t n,
    double* x, const double* l, const double* u, const mir_lsq_gpu_options* options,
    void* fContext, mir_least_squares_function_d f,
    void* gContext, mir_least_squares_jacobian_d g,
    void* tmContext, mir_least_squares_thread_manager tm);
mir_least_squares_result_s mir_optimize_least_squares_gpu_s(
    const mir_least_squares_settings_s* settings, size_t m, size_t n,
    float* x, const float* l, const float* u, const mir_lsq_gpu_options* options,
    void* fContext, mir_least_squares_function_s f,
    void* gContext, mir_least_squares_jacobian_s g,
    void* tmContext, mir_least_squares_thread_manager tm);

/* Standalone BOXCQP on the device (the reference's convenience overload QP:85-102 is D-only).
 * P: host row-major n x n, lower triangle meaningful; q,l,u,x host n-vectors. Returns BoxQPStatus
 * (mir_box_qp_numericError also when no device is usable). */
int mir_solve_box_qp_gpu_d(const mir_box_qp_settings_d* settings, size_t n, const double* P,
                           const double* q, const double* l, const double* u, double* x,
                           int unconstrainedSolution, int* iterations);
int mir_solve_box_qp_gpu_s(const mir_box_qp_settings_s* settings, size_t n, const float* P,
                           const float* q, const float* l, const float* u, float* x,
                           int unconstrainedSolution, int* iterations);

/* Batched small fits, ONE WAVEFRONT PER PROBLEM (BASELINE cfg 5): `count` independent problems of the same shape
 * with a built-in residual model r_i = model(t_i; x) - data_i evaluated inside the kernel (no callback):
 *   MIR_LSQ_MODEL_EXP_DECAY   n = 3   p0 exp(-t p1) + p2
 *   MIR_LSQ_MODEL_EXP3_AFFINE n = 8   p0 exp(-t p1) + p2 exp(-t p3) + p4 exp(-t p5) + p6 + p7 t
 * x: count x n (in/out), lower/upper: n (shared), t: m values shared by all problems (t_stride = 0) or count x m
 * (t_stride = m), data: count x m, results: count. All HOST pointers. The LM algorithm, statuses and counters are
 * those of mir_optimize_least_squares_s; problems whose step reaches a finite bound are completed by the general
 * solver (BOXCQP active set) transparently. Returns 0, or a negative value: -1 bad arguments, -2 no device, -3 a problem
 * does not fit its workgroup's LDS ((n + 2) m floats <= 160 KB - 512: m <= 4083 at n = 8, 8166 at n = 3), -4 / -5 a failed
 * allocation / launch. */
enum { MIR_LSQ_MODEL_EXP_DECAY = 0, MIR_LSQ_MODEL_EXP3_AFFINE = 1,
       MIR_LSQ_MODEL_EXP_DECAY_PAD8 = 2 };   /* n = 8: p0 exp(-t p1) + p2 + p3 sin 2t + p4 cos 2t + p5 sin 5t + p6 cos 5t + p7 t
                                                (BASELINE cfg 5: the exponential decay padded to n = 8 with terms linear in
                                                their parameters: well conditioned in fp32) */
/* Per-call options of the batched entries (nothing about them is process-wide). NULL = all defaults. */
enum { MIR_LSQ_BATCHED_ANALYTIC_JACOBIAN = 2 };   /* variant bit (launch_batched<Model> with a model that has `grad`): the model's own
                                             derivative instead of finite differences -- the reference's optional g callback:
                                             gCalls counts the refreshes, the default age limit is 3 (least_squares.d:945, 1010) */
enum { MIR_LSQ_BATCHED_NO_LADDER = 1 };   /* variant bit: every damped solve is made for ONE lambda, as the reference's loop does
                                             (boxcqp.d:194 per LS:1080); by default a solve covers lambda and the three values
                                             the rejection rule would give it next (four 16-lane groups of the wave), with
                                             the same steps, bit for bit */
typedef struct mir_lsq_batched_options {
    uint32_t struct_size;     /* = sizeof(mir_lsq_batched_options) */
    uint32_t variant;         /* MIR_LSQ_BATCHED_* bits; 0 = default */
    void* stream;             /* mir_lsq_batched_kernel_s: hipStream_t to enqueue on (NULL = the default stream) */
    float* basis;             /* optional DEVICE buffer for the model's per-row basis table (models with a basis only:
                                 (t_stride ? count : 1) x m x nb floats, 16-byte aligned), owned by the caller and filled by
                                 every call. NULL: the call allocates the table (hipMalloc) and synchronises the stream before
                                 freeing it -- the ONLY case in which mir_lsq_batched_kernel_s synchronises; pass a table to
                                 stay asynchronous (bench.py does) */
    size_t basis_bytes;       /* size of `basis` (the call fails with -1 when it is too small) */
    uint64_t* timing;         /* profiling builds only (-DMIRLSQ_BATCHED_TIMING): DEVICE buffer of 10 cycle counters per
                                 problem, written by the kernel; ignored otherwise */
} mir_lsq_batched_options;

int mir_optimize_least_squares_batched_s(const mir_least_squares_settings_s* settings, size_t count, size_t m, int model,
                                         float* x, const float* lower, const float* upper,
                                         const float* t, size_t t_stride, const float* data,
                                         mir_least_squares_result_s* results, const mir_lsq_batched_options* options);

/* The same wave-per-problem kernel on DEVICE-RESIDENT data (every pointer is a device pointer; `results` receives
 * `count` records in place; enqueued on options->stream, no synchronisation -- see mir_lsq_batched_options.basis for the one
 * exception): what bench.py --config cfg5 times. Problems whose step reaches a finite bound come back with status -100
 * (MIR_LSQ_BATCHED_NEEDS_GENERAL): the host entry above completes those with the general solver, this one leaves that to
 * the caller. Returns 0 when the launch succeeded.
 * A caller with a residual model of its own -- the reference takes an arbitrary f, least_squares.d:73-80 -- compiles the
 * same kernel for it from include/mir_optim_amd_batched.hpp (launch_batched<Model>); the three built-in models are
 * instances of that template. */
enum { MIR_LSQ_BATCHED_NEEDS_GENERAL = -100 };
int mir_lsq_batched_kernel_s(const mir_least_squares_settings_s* settings, size_t count, size_t m, int model,
                             float* x, const float* lower, const float* upper,
                             const float* t, size_t t_stride, const float* data,
                             mir_least_squares_result_s* results, const mir_lsq_batched_options* options);

/* Resident-J solver (include/mir_optim_amd_resident.hpp, launch_resident<Model>): the whole loop of least_squares.d:972-1175
 * in ONE cooperative launch for problems whose Jacobian, residuals and per-row data fit the LDS of the chip (BASELINE
 * cfg 2). The residual model is a compile-time type of the caller's, as on the batched path. Options and statistics of
 * that launch; every pointer is a DEVICE pointer. */
enum { MIR_LSQ_RESIDENT_NO_NULL_SKIP = 1u,     /* variant bit: evaluate f also for trials equal to x bit for bit */
       MIR_LSQ_RESIDENT_UNBOUNDED = 2u,         /* variant bit: the caller asserts that every lower / upper entry is infinite
                                                  (the BOXCQP active-set loop is compiled out of workgroup 0's solve) */
       MIR_LSQ_RESIDENT_NO_LOOKAHEAD = 4u,      /* variant bit: every trial gets a round of its own (no sums of squares of the
                                                  next damping levels evaluated along; same results bit for bit, for A/B runs) */
       MIR_LSQ_RESIDENT_DEBUG_DROP_WORKGROUP = 16u, /* diagnostic (tests): the last workgroup leaves before the first round; the
                                                  others' waits are bounded (1 s in this mode, 20 s otherwise) and the launch
                                                  must end with numericError and mir_lsq_resident_stats.abort_code != 0 */
       MIR_LSQ_RESIDENT_ANALYTIC_JACOBIAN = 32u, /* the model's own derivative (Model::jac) instead of finite differences at every
                                                  refresh: the reference's optional g callback (gCalls, age limit 3 by default).
                                                  launch_resident answers -1 for a model without jac */
       MIR_LSQ_RESIDENT_NO_STAMPS = 8u          /* variant bit: mir_lsq_resident_stats carries the counters only, every t_* is 0 (the
                                                  clock reads of workgroup 0 cost a few per cent of a latency-bound fit) */ };
typedef struct mir_lsq_resident_stats {         /* written by the kernel at exit; times in 10 ns ticks of workgroup 0 */
    uint64_t rounds, passes, accepted, rejected, step_guard_rejects, jacobian_full, jacobian_broyden, qp_active_set_passes,
        elided_evaluations;
    uint64_t t_total, t_stage, t_worker, t_group, t_total_wait, t_solver, t_solve_body, t_cmd_wait;
    uint64_t t_w_eval, t_w_fd, t_w_prod;         /* of t_worker: trial residuals, finite-difference refreshes, products + publication */
    uint64_t t_w_mma;                            /* of t_w_prod: the matrix-core loop up to the workgroup's cross-wave hand-over */
    uint64_t lookahead_rejections;               /* rejected passes decided from a sum of squares evaluated along an earlier round */
    uint64_t t_look;                             /* of t_solver: those decisions (own rows + collecting the others' sums) */
    uint64_t t_unpack, t_publish;                /* of t_solver: totals of the 16 groups -> LDS; the next command (constants, stores, drain) */
    uint32_t abort_code, grid, rows, groups;
} mir_lsq_resident_stats;
typedef struct mir_lsq_resident_options {
    uint32_t struct_size;            /* = sizeof(mir_lsq_resident_options) */
    uint32_t variant;                /* MIR_LSQ_RESIDENT_* bits */
    void* stream;                    /* hipStream_t to enqueue on (NULL = the default stream) */
    void* workspace;                 /* optional device scratch of >= resident_workspace_bytes<Model>(m) bytes, owned by the
                                        caller; NULL: the call allocates it and synchronises the stream before freeing it */
    size_t workspace_bytes;
    mir_lsq_trace_record* trace_records;   /* optional per-pass trace (same events as mir_lsq_trace) */
    uint32_t trace_capacity;
    uint32_t max_workgroups;         /* 0 = one workgroup per CU of the device */
    uint32_t* trace_count;           /* events of the solve (also beyond the capacity) */
    mir_lsq_resident_stats* stats;   /* optional */
} mir_lsq_resident_options;

/* Unit-level access to the damped solve of that kernel (?posvx('E','L') with one matrix row per lane; what boxcqp.d:194
 * calls): `count` systems of order n (3 or 8, the orders of the compiled-in models), device pointers; P count x 64 floats,
 * system p at P + 64 p, row-major with row stride 8, the LOWER triangle is read; rhs and x count x 8 floats (components
 * >= n ignored / zero); info[p] = 0 or the order of the leading minor that is not positive (x of that system is zero).
 * Enqueued on `stream`, no synchronisation. */
int mir_lsq_batched_posvx_s(size_t count, size_t n, const float* P, const float* rhs, float* x, int* info, void* stream);

/* Unit-level access to the hot kernels (parity tests and micro-benchmarks). All pointers are
 * DEVICE pointers; stream may be NULL (default stream; the call synchronises before returning).
 * JJ: n x n row-major, full symmetric on return. broyden != 0 first applies
 *   J += ((y - y_old - J dx) / (dx.dx)) dx^T  (LS:1002-1006) in the same pass. Returns 0 on success. */
int mir_lsq_jtj_d(size_t m, size_t n, double* J, const double* y, const double* y_old, const double* dx,
                  int broyden, double* JJ, double* Jy, void* stream, float* kernel_ms);
int mir_lsq_jtj_s(size_t m, size_t n, float* J, const float* y, const float* y_old, const float* dx,
                  int broyden, float* JJ, float* Jy, void* stream, float* kernel_ms);
/* Finite-difference fill fused into the J^T J kernel (f64; n <= 256; else -6). Yrm: m x 2n
 * row-major, Yrm[i][2j] = f(x + h e_j)_i, Yrm[i][2j+1] = f(x - h e_j)_i; twh[j] = (x_j + h) - (x_j - h) after clipping
 * (0 = collapsed interval: zero column, LS:1046). Writes J (m x n row-major, (Y+ - Y-) * (1 / twh) as LS:1041-1047),
 * JJ = J^T J (full symmetric) and Jy = J^T y. */
int mir_lsq_fd_jtj_d(size_t m, size_t n, const double* Yrm, const double* twh, const double* y, double* J,
                     double* JJ, double* Jy, void* stream, float* kernel_ms);
/* The same from the m x n row-major DIFFERENCE panel D[i][j] = f(x + h e_j)_i - f(x - h e_j)_i (what a fbRowMajorDiff callback
 * writes; f64; n <= 128, n = 192, n = 256, else -6): J = D * (1 / twh) column-wise (zero for twh = 0), JJ, Jy. */
int mir_lsq_fd_diff_jtj_d(size_t m, size_t n, const double* Drm, const double* twh, const double* y, double* J,
                          double* JJ, double* Jy, void* stream, float* kernel_ms);

/* Workspace: device buffers for one (m, n, element size) problem, reusable across calls. */
mir_lsq_workspace* mir_lsq_workspace_create(size_t m, size_t n, size_t elem_size);
void mir_lsq_workspace_destroy(mir_lsq_workspace* ws);

/* Row-shard communicators: one fused sum-all-reduce of the packed
 * [J^T J lower | J^T y] buffer per Jacobian-changing pass, one 1-scalar all-reduce per trial step. */
int mir_lsq_rccl_unique_id(void* out_128_bytes);                       /* ncclGetUniqueId */
mir_lsq_comm* mir_lsq_comm_create_rccl(int nranks, int rank, const void* unique_id_128_bytes);
/* callback communicator: `allreduce(ctx, device_buf, count_of_doubles, stream)` must sum `buf` over
 * ranks in place, ordered after prior work on `stream` (used with gloo in the tests). */
typedef void (*mir_lsq_allreduce_fn)(void* ctx, double* device_buf, size_t count, void* stream);
mir_lsq_comm* mir_lsq_comm_create_callback(int nranks, int rank, mir_lsq_allreduce_fn fn, void* ctx);
/* In-process group: `nranks` solver instances inside ONE process -- one host thread each, on the same device (R logical
 * shards on one GPU: the single-device emulation of SURVEY.md section 7 step 7) or on different devices (one process
 * driving several GPUs without RCCL). An all-reduce copies each rank's buffer to pinned host memory, the ranks meet at a
 * barrier, every rank sums the nranks contributions in rank order (bitwise identical totals on all ranks) and copies the
 * total back. out_comms receives nranks handles (rank r at out_comms[r]); destroy each with mir_lsq_comm_destroy -- in any
 * order and from any thread, also while other ranks are still solving: the shared slots live until the last handle goes.
 * A rank that waits longer than 120 s at the barrier gives up (the solve then returns numericError). Returns 0. */
int mir_lsq_comm_create_local_group(int nranks, mir_lsq_comm** out_comms);
void mir_lsq_comm_destroy(mir_lsq_comm* comm);
/* Record / replay of a rank's exchanges (a measurement tool: bench.py --replay-ranks). A rank of an
 * in-process group can RECORD the totals of all its all-reduces (as doubles, concatenated in call order) into a caller-owned
 * host buffer: mir_lsq_comm_record(comm, buf, capacity_in_doubles) before the solve, mir_lsq_comm_recorded(comm) after it
 * (the doubles written; (size_t)-1 after an overflow). A REPLAY communicator then lets ONE rank run alone on exactly the global
 * trajectory: every all-reduce REPLACES the buffer by the next recorded total (a stream-ordered device copy; the rank's own
 * contribution is computed and discarded) and, when `inner` is given, passes it through inner's all-reduce as well (a
 * one-rank RCCL communicator: the launch cost of the real collective). mir_lsq_comm_replay_rewind restarts the tape for the
 * next solve. f64 problems only. The replay handle does not own `inner`. */
int mir_lsq_comm_record(mir_lsq_comm* comm, double* host_buf, size_t capacity);
size_t mir_lsq_comm_recorded(const mir_lsq_comm* comm);
mir_lsq_comm* mir_lsq_comm_create_replay(int nranks, int rank, const double* totals_host, size_t len, mir_lsq_comm* inner);
int mir_lsq_comm_replay_rewind(mir_lsq_comm* comm);
/* A MODEL of the collectives' latency for the replay tool: every replayed exchange first holds the stream for `microseconds`
 * (one idle wave), so that a one-GPU replay of a rank charges each all-reduce what an N-rank RCCL call is ASSUMED to cost --
 * bench.py --replay-ranks R --replay-latency-us L prints the solve time as a function of that assumption. 0 = off. */
int mir_lsq_comm_replay_set_delay(mir_lsq_comm* comm, unsigned microseconds);
/* ranks of the communicator as its transport reports them (RCCL: ncclCommCount); -1 on error */
int mir_lsq_comm_ranks(const mir_lsq_comm* comm);
/* One line about the transport, for logs: "rccl path=<shared object ncclAllReduce was bound from> version=<ncclGetVersion>
 * preloaded=<1: the host program had it mapped already> ranks=<ncclCommCount> rank=<r>", or "callback ..." / "local-group ...".
 * snprintf semantics (returns the length it needs). bench.py puts it into its JSON line: the library binds an RCCL the
 * host program has already loaded (PyTorch's) in preference to /opt/rocm's, and a version skew should be visible. */
int mir_lsq_comm_describe(const mir_lsq_comm* comm, char* buf, size_t len);
/* Sum `count` doubles (floats) of the DEVICE buffer `buf` over the communicator's ranks, in place, ordered on `stream` -- the
 * collective the solver issues for the three exchanges of a pass (least_squares.d:1052, 1065, 1115), exposed so that a caller
 * can check a communicator before the first solve (bench.py does). Returns 0 on success. */
int mir_lsq_comm_allreduce_d(mir_lsq_comm* comm, double* buf, size_t count, void* stream);
int mir_lsq_comm_allreduce_s(mir_lsq_comm* comm, float* buf, size_t count, void* stream);

/* Small device utilities for language bindings that have no HIP runtime of their own. */
int mir_lsq_device_count(void);
void* mir_lsq_device_malloc(size_t bytes);
void mir_lsq_device_free(void* p);
int mir_lsq_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes, void* stream);
int mir_lsq_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes, void* stream);
int mir_lsq_memcpy_d2d(void* dst_device, const void* src_device, size_t bytes, void* stream);   /* asynchronous */

/* Self-test of the wave reductions every kernel ends with (csrc/common.h: wave_sum / wave_max on DPP row operations and
 * v_readlane): 2048 x 256 threads compare them, bit for bit, with the plain butterfly on `rounds` pseudo-random inputs a lane.
 * mismatches[0..3] = lanes that differ for sum<float>, sum<double>, max<float>, max<double> (all 0 on a healthy device).
 * Returns 0 when the test ran. */
int mir_lsq_selftest_reductions(int rounds, int mismatches[4]);
void* mir_lsq_stream_create(void);
void mir_lsq_stream_destroy(void* stream);
int mir_lsq_stream_synchronize(void* stream);
/* "mir_optim_amd <major.minor> (gfx950)". 0.2: the last argument of mir_lsq_batched_kernel_s became the options struct (it was a
 * hipStream_t in 0.1 -- same arity, so an old caller still links: both batched entries answer -1 to an options pointer whose
 * first word is not a plausible struct_size) and mir_optimize_least_squares_batched_s gained it; 0.3: the resident-J options /
 * statistics of part 2 and mir_optim_amd_resident.hpp. Callers that cache function pointers across versions check this. */
const char* mir_lsq_version(void);

/* =====================================================================================================
 * Part 3 -- a caller of the path: fitSpline (/root/reference/source/mir/optim/fit_splie.d:26-85).
 *
 * Fits the VALUES of a cubic spline at fixed knots x[nx] to scattered points by least squares through the
 * library's own LM entry (n = nx, m = npoints + (lambda == 0)). The spline is mir.interpolate.spline's default
 * configuration (mir-algorithm, un-vendored): C2 cubic, not-a-knot ends, kept in Hermite form (values + first
 * derivatives). Quirks of the reference kept on purpose (its unittest FS:88-141 pins them):
 *   - the smoothness penalty integrates the spline's FIRST derivative at the knots (FS:74, FS:77 index [1]);
 *   - y[m-1] is always overwritten by the penalty term (FS:83): with lambda != 0 the last point is ignored.
 * points: npoints x 2 row-major {abscissa, value}; l, u: bounds on the spline values; dist: optional replacement of
 * the reference's `alias d = "a - b"` (NULL = a - b); splineY (out, nx): fitted values, started from 0 (FS:56-57);
 * splineD (out, nx, optional): first derivatives. Returns MIR_FIT_SPLINE_*; the LM status is in *result (the D
 * function throws for result->status < 0 through `optimize`, LS:175-179, and for TOO_FEW_POINTS, FS:47-51).
 * ===================================================================================================== */
enum { MIR_FIT_SPLINE_OK = 0, MIR_FIT_SPLINE_TOO_FEW_POINTS = 1, MIR_FIT_SPLINE_BAD_ARGUMENT = 2 };

int mir_fit_spline_d(const mir_least_squares_settings_d* settings, size_t npoints, const double* points, size_t nx,
                     const double* x, const double* l, const double* u, double lambda, double (*dist)(double, double),
                     double* splineY, double* splineD, mir_least_squares_result_d* result);
int mir_fit_spline_s(const mir_least_squares_settings_s* settings, size_t npoints, const float* points, size_t nx,
                     const float* x, const float* l, const float* u, float lambda, float (*dist)(float, float),
                     float* splineY, float* splineD, mir_least_squares_result_s* result);
/* the residual function fitSpline minimises (FS:60-84), exposed so that tests can hand the same function to the oracle */
void mir_fit_spline_residuals_d(size_t npoints, const double* points, size_t nx, const double* x, double lambda,
                                const double* splineY, size_t m, double* y);
/* C2 not-a-knot cubic spline: first derivatives at the knots; value / 1st / 2nd derivative at t (out3[3]) */
void mir_spline_c2_derivatives_d(size_t n, const double* x, const double* y, double* d);
void mir_spline_c2_derivatives_s(size_t n, const float* x, const float* y, float* d);
void mir_spline_eval_d(size_t n, const double* x, const double* y, const double* d, double t, double* out3);
void mir_spline_eval_s(size_t n, const float* x, const float* y, const float* d, float t, float* out3);

#ifdef __cplusplus
}
#endif
#endif /* MIR_OPTIM_AMD_H */
