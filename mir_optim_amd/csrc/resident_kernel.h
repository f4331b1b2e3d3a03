// resident_kernel.h -- the WHOLE Levenberg-Marquardt loop of optimizeLeastSquaresImplGeneric!T
// (/root/reference/source/mir/optim/least_squares.d:877-1176, cited LS:nnn) in ONE cooperative launch, for problems whose
// Jacobian fits the LDS of the chip (BASELINE cfg 2: m = 1e5 x n = 16, J = 12.8 MB against 256 CUs x 160 KB). SURVEY 7(e).
//
//   * one workgroup per CU; workgroup w keeps ITS row slice of J (row-major, LS:918), of the residual y, the trial residual
//     and the caller's per-row data in LDS for the whole solve -- after the data is staged once, no pass touches HBM;
//   * the residual model is a compile-time type (include/mir_optim_amd_resident.hpp), evaluated one row per thread:
//     finite-difference refresh (LS:1018-1049) and Broyden update (LS:1002-1006) happen in place, in LDS;
//   * J^T J / J^T y of a slice (LS:1052, 1065) on v_mfma_f64_16x16x4 straight from the LDS rows; per-workgroup partials are
//     summed in a FIXED order through L2: members -> 16 group leaders -> workgroup 0 (bitwise reproducible, no float atomics);
//   * workgroup 0 runs the n x n part of the pass -- lambda_0, P = J^T J + lambda I, solveBoxQP with ?posvx('E','L') and the
//     BOXCQP active-set loop (boxcqp.d:122-379), step rounding, trial point, prediction: lm_solve_body, solve_kernel.h, the
//     code of the launch-chain path -- and the scalar logic of the loop (acceptance, lambda / mu schedule, ageing,
//     convergence tests: LS:972-1175) and publishes ONE command per round; every workgroup executes it on its rows;
//   * a round = [workers: commit / discard the previous trial; evaluate f(trial) on their rows; SPECULATIVELY form the Broyden
//     update that an acceptance of this trial would be followed by and contract J'^T J', J'^T y' from it] -> reduce ->
//     [workgroup 0: decision, next solve] -> command. One round per executed pass of the reference's loop (a refresh adds one),
//     three hand-offs per round, no grid-wide barrier. A rejected trial discards the speculative products (J is only
//     rewritten by the commit of the NEXT command), so the trajectory is the reference's, pass for pass.
//
// Hand-offs between workgroups follow the sc1 recipe of the CDNA4 guide: every shared word is written and read with agent-scope
// relaxed atomics (write-through stores, L1-bypassing loads), the storing waves drain (s_waitcnt vmcnt(0)) before a workgroup
// barrier, then ONE lane signals (counter add or flag store); the consumer polls ONE word, passes a workgroup barrier and
// loads. Counters and flags are monotonic within a launch (round numbers) and zeroed by a memset node before it. Every spin
// is bounded (kResSpinSeconds): a workgroup that gives up raises `abort`, everybody leaves, the result is numericError.
#pragma once

#include "../../include/mir_optim_amd.h"
#include "common.h"
#include "solve_types.h"
#include "solve_kernel.h"
#include "solve_wave16.h"

namespace mirlsq {

// Threads of a workgroup. 256: what lm_solve_body (n > 16) is written for. The kernel also runs with 512 for n <= 16 (the one-wave
// solve does not care, and two waves a SIMD keep the f64 vector pipe busier in the one-row-per-thread phases: measured at cfg 2,
// trial residuals + products + refreshes 524 -> 392 us per fit) -- but two waves a SIMD leave a wave 256 registers where one has
// 512, the solve then spills 150-240 of them into the round loop and workgroup 0's share grows by more (384 -> 700 us; with the
// solve out of line -- __noinline__, operands in LDS -- the workers go 338 -> 289 us and workgroup 0 396 -> 908): 256 it is.
__host__ __device__ constexpr int res_threads(int n) { return n <= 16 ? 256 : kSolveThreads; }
constexpr int kResGroups = 16;                  // group leaders (first level of the reduction)
constexpr int kResGroupMax = 16;                // members a leader sums (grid <= 256)
constexpr int kResNMax = 32;                    // parameters: one or two 16-column blocks (the solve runs NB = 1 or 2)
constexpr double kResSpinSeconds = 20.0;

enum : uint32_t { kResEval = 1, kResEvalSpec = 2, kResFd = 3, kResExit = 4 };          // command actions
enum : uint32_t { kResPreAccept = 1, kResPreCommitJ = 2 };                             // what to do with the previous trial first
enum : uint32_t { kResVariantNoNullSkip = MIR_LSQ_RESIDENT_NO_NULL_SKIP, kResVariantUnbounded = MIR_LSQ_RESIDENT_UNBOUNDED,
                  kResVariantNoLookahead = MIR_LSQ_RESIDENT_NO_LOOKAHEAD, kResVariantNoStamps = MIR_LSQ_RESIDENT_NO_STAMPS,
                  kResVariantDebugDrop = MIR_LSQ_RESIDENT_DEBUG_DROP_WORKGROUP, kResVariantAnalytic = MIR_LSQ_RESIDENT_ANALYTIC_JACOBIAN };

// a model MAY provide the analytic Jacobian of a row -- the reference's optional g callback (least_squares.d:80, 1010-1014):
//     __device__ static void jac(const double* row, const double* c, double* Ji);     // Ji[0 .. n): d residual_i / d x_j
template <class Model, class = void> struct res_has_jac : std::false_type {};
template <class Model>
struct res_has_jac<Model, std::void_t<decltype(Model::jac((const double*)nullptr, (const double*)nullptr, (double*)nullptr))>> : std::true_type {};

using ResidentStats = mir_lsq_resident_stats;     // written by workgroup 0 at exit (times: 10 ns ticks)

// payload of one workgroup's contribution: [ sum of squares | J^T y (NC) | lower block triangle of J^T J: diagonal blocks packed
// (136), the others whole (256) ]
template <int NCB> struct ResPayload {
    static constexpr int NC = 16 * NCB;
    static constexpr int NBT = NCB * (NCB + 1) / 2;
    static constexpr int JY = 1;
    static constexpr int JJ = 1 + NC;
    __host__ __device__ static constexpr int blk_base(int b)
    {
        int off = JJ, I = 0, J = 0;
        for (int k = 0; k < b; ++k) { off += (I == J) ? 136 : 256; if (J == I) { ++I; J = 0; } else ++J; }
        return off;
    }
    static constexpr int LEN = blk_base(NBT);
    static constexpr int STRIDE = (LEN + 15) / 16 * 16;
};

constexpr int kResCMax = 64;                     // per-point constants of a model (Model::nc)
constexpr int kResCmdWords = 2 * kResNMax + 2 + kResCMax;   // point | dx | 1 / dx.dx | action + preops << 32 | the point's constants
// Look-ahead (n <= 16, section "look-ahead" in the kernel): behind the command, the constants of up to kResLookMax FURTHER ladder
// levels' trial points; bits 16-17 of the action word say how many. Workers evaluate their sums of squares AFTER they have
// published the round's payload -- while the leaders and workgroup 0 are busy and they would wait -- into look[wg][level].
constexpr int kResLookMax = 3;
constexpr int kResCmdWordsAll = kResCmdWords + kResLookMax * kResCMax;

struct ResidentArgs {
    LmSettingsDev<double> set;
    uint32_t maxIterations, maxAge, variant;
    int m, grid, rows, groups;          // rows: slice length per workgroup (ceil(m / grid)); groups = min(kResGroups, grid)
    const double* rowdata;              // m x Model::nd
    double* x;                          // n, in / out
    const double* lower;
    const double* upper;
    mir_least_squares_result_d* result;
    // workspace (zeroed: cnt, flag, seq, abort)
    double* partial;                    // grid x STRIDE
    double* gtotal;                     // groups x STRIDE
    uint32_t* cnt;                      // groups counters, 32 words apart
    uint32_t* flag;                     // groups flags, 32 words apart
    uint32_t* seq;                      // command sequence number
    uint32_t* abort;
    uint32_t* lcnt;                     // look-ahead sums published (one add per workgroup and round that carries any)
    double* look;                       // grid x 4: the workgroups' look-ahead sums of squares
    unsigned long long* cmd;            // kResCmdWordsAll
    double* JJ[2];                      // n x n each: current / speculative
    double* Jy[2];
    double* xs;                         // n: the solver's x
    double* dx;                         // n
    double* trial;                      // n
    LmState<double>* st;
    ChainRec<double>* rec;
    SolveScratch<double> sc;
    mir_lsq_trace_record* trace;        // optional device buffer
    uint32_t trace_capacity;
    uint32_t* trace_count;
    ResidentStats* stats;               // optional
};

// ---- agent-scope relaxed accesses: sc1 stores / loads (write-through, L1-bypassing)
__device__ __forceinline__ void res_st(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double res_ld(const double* p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void res_st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long res_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void res_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t res_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void res_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ONE wave (wave 0) waits until *p >= target; the caller puts a workgroup barrier behind it. Returns false after
// kResSpinSeconds or when another workgroup has raised `abort`.
__device__ __forceinline__ bool res_wait_ge(const uint32_t* p, uint32_t target, const uint32_t* abort, double seconds = kResSpinSeconds)
{
    long long t0 = 0;
    for (uint32_t spins = 0;; ++spins) {
        if (res_ld(p) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u) {
            if (res_ld(abort) != 0) return false;
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if ((double)(now - t0) > seconds * 1e8) return false;
        }
    }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// Model (include/mir_optim_amd_resident.hpp): n, nd, nc, prepare(x, c), eval(row, c).
template <class Model, bool BOUNDED>
__global__ __launch_bounds__(res_threads(Model::n)) void k_lm_resident(ResidentArgs a)
{
    constexpr int kResThreads = res_threads(Model::n), kResWaves = kResThreads / kWave;
    constexpr int N = Model::n, ND = Model::nd, NCN = Model::nc;
    static_assert(N >= 1 && N <= kResNMax, "1 <= n <= 32");
    static_assert(NCN >= 1 && NCN <= kResCMax, "1 <= nc <= 64");
    constexpr int NCB = (N + 15) / 16;
    constexpr int NC = 16 * NCB;
    constexpr int NB = NCB;                                 // solve_lds.h block count: 1 (n <= 16) or 2
    using PL = ResPayload<NCB>;
    using Acc = typename Mma<double>::Acc;
    constexpr int NBT = PL::NBT;
    constexpr int REDW = NBT * 256 + NC;                    // one wave's block accumulators + J^T y

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_res[];
    __shared__ int s_ok;
    __shared__ double s_w[kResWaves];
    __shared__ unsigned long long s_cmd[kResCmdWords];
    __shared__ double s_lw[kResLookMax][kResWaves];         // look-ahead: the waves' sums of squares per level
    __shared__ double s_look[kResGroups + 4];               // ... group sums | [kResGroups + k]: this workgroup's level sums, the total

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x;
    auto wave_total = [&](const double* w) {
        double t = (w[0] + w[1]) + (w[2] + w[3]);
        if constexpr (kResWaves == 8) t += (w[4] + w[5]) + (w[6] + w[7]);
        return t;
    };
    constexpr int RPAD = 8 * kResWaves;                     // every wave takes two MFMA steps of four rows per iteration
    const int R = (a.rows + RPAD - 1) / RPAD * RPAD;        // padded slice
    const int row0 = wg * a.rows;
    const int nrows = max(0, min(a.rows, a.m - row0));
    const int G = a.grid, NG = a.groups;
    const int grp = wg % NG;
    const bool leader = wg < NG;
    const int members = (G - grp + NG - 1) / NG;            // workgroups w < G with w % NG == grp

    // J rows have stride JS = NC + 1 doubles: a thread walking ITS row (the Broyden dot product) and a wave reading 4 rows x 16
    // columns (the matrix-core operand) are both free of bank conflicts (17 doubles = 34 banks, 33 = 66)
    constexpr int JS = NC + 1;
    double* Jl = reinterpret_cast<double*>(smem_res);       // R x JS
    double* Yb = Jl + (size_t)R * JS;                       // 3 x R: residual, trial residual, Broyden scale u
    double* Dl = Yb + 3 * (size_t)R;                        // R x ND
    double* Cl = Dl + (size_t)R * ND;                       // 2 N x NCN point constants
    double* RED = Cl + 2 * N * NCN;                         // kResWaves x REDW ; later the totals (PL::LEN)
    double* Xl = RED + kResWaves * REDW;                    // NC: point of an FD refresh
    double* DXl = Xl + NC;                                  // NC: dx of the speculative update (kept for the commit)
    double* INVl = DXl + NC;                                // NC: 1 / (x+ - x-) per column
    // n <= 16: the n x n part runs on ONE wave with a matrix row per lane (solve_wave16.h) and everything it reads and writes
    // stays in LDS; above, lm_solve_body (the launch-chain path's code, 256 threads, operands and scratch in global memory)
    constexpr bool WAVE = N <= kW16;
    double* SOL = INVl + NC;
    unsigned char* solve_smem = reinterpret_cast<unsigned char*>(SOL);
    // WAVE layout of SOL: J^T J x 2 (16 x 16 each) | J^T y x 2 | x | lower | upper | 4 ladder levels: dx, trial | 4 level records
    double* JJp[2] = {WAVE ? SOL : a.JJ[0], WAVE ? SOL + 256 : a.JJ[1]};
    double* Jyp[2] = {WAVE ? SOL + 512 : a.Jy[0], WAVE ? SOL + 528 : a.Jy[1]};
    double* xsp = WAVE ? SOL + 544 : a.xs;
    double* lop = SOL + 560;
    double* upp = SOL + 576;
    double* dxp = WAVE ? SOL + 592 : a.dx;                   // level k at dxp + 16 k
    double* trp = WAVE ? SOL + 656 : a.trial;
    double* lrec = SOL + 720;                                // level k: lambda, ndd, pred, xnorm at lrec + 8 k; ints behind
    int* lreci = reinterpret_cast<int*>(SOL + 752);          // level k: qp_status, qp_iters, flags, offered at lreci + 4 k
    constexpr int JLD = WAVE ? kW16 : N;                     // leading dimension of the J^T J copies
    // look-ahead needs the ladder of the one-wave solve, room for 1 + kResLookMax constant sets in Cl (2 N of them) and one
    // thread per constant when the command is read
    constexpr bool LOOK = WAVE && N >= 2 && kResLookMax * NCN <= kResThreads;

    int iy = 0, it = 1;                                     // roles of Yb's first two vectors
    constexpr int iu = 2;
    long long tk0 = 0, t_stage = 0, t_w_eval = 0, t_w_fd = 0, t_w_prod = 0, t_w_mma = 0, t_worker = 0, t_group = 0, t_total_wait = 0, t_solver = 0, t_solve_body = 0, t_cmd_wait = 0;
    const bool stamper = wg == 0 && tid == 0;
    const bool clk = stamper && a.stats && !(a.variant & kResVariantNoStamps);     // s_memrealtime is not free: ~20 reads a round
    if (stamper) tk0 = wall_clock64();

    // ---- stage the slice's row data; zero J, the vectors and the padding
    for (int e = tid; e < R * JS; e += kResThreads) Jl[e] = 0;
    for (int e = tid; e < 3 * R; e += kResThreads) Yb[e] = 0;
    for (int e = tid; e < R * ND; e += kResThreads) {
        const int i = e / ND;
        Dl[e] = i < nrows ? a.rowdata[(size_t)row0 * ND + e] : 0.0;
    }
    if (tid < NC) { DXl[tid] = 0; INVl[tid] = 0; Xl[tid] = 0; }
    if constexpr (WAVE) {
        if (wg == 0) {
            for (int e = tid; e < 768; e += kResThreads) SOL[e] = 0;
            __syncthreads();
            if (tid < kW16) {
                lop[tid] = tid < N ? a.lower[tid] : -Lim<double>::inf();
                upp[tid] = tid < N ? a.upper[tid] : Lim<double>::inf();
            }
        }
    }
    __syncthreads();
    if (clk) t_stage = wall_clock64() - tk0;
    // diagnostic (tests): the last workgroup leaves before the first round -- everybody else must give up (bounded spins, 1 s in
    // this mode) and the launch must end with numericError + an abort code, not hang
    const double spin_s = (a.variant & kResVariantDebugDrop) ? 1.0 : kResSpinSeconds;
    if ((a.variant & kResVariantDebugDrop) && a.grid > 1 && wg == a.grid - 1) return;

    // ---- the command being executed (the first one comes from the arguments: evaluate f at x0)
    uint32_t action = kResEval, preops = 0;
    double inv_dd = 0;
    if (tid < NC) Xl[tid] = tid < N ? a.x[tid] : 0.0;       // point of the command (Xl doubles as that)
    __syncthreads();
    uint32_t round = 0;

    // ---- solver state (workgroup 0; every thread carries the same values)
    double lambda = 0, mu = 1, residual = 0, dx_dot = 0, s_ndd = 0, s_pred = 0, s_xnorm = 0, s_lam_used = 0;
    uint32_t age = 0, iterations = 0, fCalls = 0;
    // MIR_LSQ_RESIDENT_ANALYTIC_JACOBIAN: refreshes call Model::jac instead of differencing (g of LS:1010-1014: gCalls, and the
    // default age limit of a caller WITH a Jacobian, LS:945)
    constexpr bool HAS_JAC = res_has_jac<Model>::value;
    const bool use_g = HAS_JAC && (a.variant & kResVariantAnalytic) != 0;
    uint32_t gCalls = 0;
    const uint32_t maxAge = a.maxAge ? a.maxAge : (use_g ? 3u : 2u * N);   // LS:945
    int status = -1;                                        // maxIterations, LS:971
    bool needJac = true, fConverged = false, x_nan = false;
    int cur = 0;                                            // which of JJ[2] / Jy[2] is the current pair
    double null_lambda = Lim<double>::inf();                // above this damping the rounded step is provably zero (see kSolve)
    int null_lambda_for = 0;                                // 2: null_lambda belongs to the current (J^T y, x)
    bool lad_valid = false;                                 // n <= 16: the ladder in LDS was solved on the current J^T J, J^T y, x
    int lad_level = 0;                                      // its level the pass being decided uses
    int phase = 0;                                          // 0: initial residual, 1: refresh products, 2: trial
    uint32_t nlook = 0;                                     // everybody: look-ahead levels of the command being executed
    int look_n = 0, look_base = 0;                          // workgroup 0: levels look_base + 1 ... + look_n of the ladder were sent along
    uint32_t look_target = 0;                               // ... value of *lcnt when every other workgroup has published them
    bool look_fetched = false;
    uint64_t n_rounds = 0, n_passes = 0, n_acc = 0, n_rej = 0, n_guard = 0, n_fd = 0, n_br = 0, n_qp = 0, n_elided = 0, n_look = 0;
    long long t_look = 0, t_unpack = 0, t_publish = 0;
    uint32_t tr_count = 0;
    auto trace = [&](int ev, uint32_t iters, double lam, double res, double tres, double dd) {
        if (tid == 0 && a.trace) {
            if (tr_count < a.trace_capacity) {
                mir_lsq_trace_record r;
                r.event = ev; r.iterations = iters; r.lambda = lam; r.residual = res; r.trial_residual = tres; r.dx_dot = dd;
                a.trace[tr_count] = r;
            }
        }
        ++tr_count;
    };
    auto fail_out = [&](uint32_t code) {                    // a spin gave up: everybody leaves
        if (tid == 0) { res_st(a.abort, code); }
        if (wg == 0 && tid == 0) {
            mir_least_squares_result_d r;
            r.status = mir_ls_numericError; r.iterations = iterations; r.fCalls = fCalls; r.gCalls = 0;
            r.residual = residual; r.lambda = lambda;
            *a.result = r;
            if (a.stats) a.stats->abort_code = code;
        }
    };

    // Sums of squares of f at `count` further points (constant sets first, first + 1, ... of Cl) over the slice's rows: the SAME
    // per-thread order, wave sum and wave order as the trial evaluation of step (2), so a level's sum has the bits a round of its
    // own would produce. Result in s_look[kResGroups + k], k < count.
    auto look_sums = [&](int first, int count) {
        double q0 = 0, q1 = 0, q2 = 0;
        for (int i = tid; i < R; i += kResThreads) {
            if (i < nrows) {                                                   // (padding rows add fma(0, 0, s) = s)
                const double* row = Dl + (size_t)i * ND;
                const double v0 = Model::eval(row, Cl + (size_t)first * NCN);
                q0 = fma(v0, v0, q0);
                if (count > 1) { const double v1 = Model::eval(row, Cl + (size_t)(first + 1) * NCN); q1 = fma(v1, v1, q1); }
                if (count > 2) { const double v2 = Model::eval(row, Cl + (size_t)(first + 2) * NCN); q2 = fma(v2, v2, q2); }
            }
        }
        q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2);
        if (lane == 0) { s_lw[0][wave] = q0; s_lw[1][wave] = q1; s_lw[2][wave] = q2; }
        __syncthreads();
        if (tid < count) s_look[kResGroups + tid] = wave_total(s_lw[tid]);
        __syncthreads();
    };

    for (;;) {
        ++round;
        long long tw0 = 0;
        if (clk) tw0 = wall_clock64();
        // =================================================================================== worker part
        // (1) what became of the previous trial
        // (the commit of a speculative Broyden step, J += u dx^T, was applied when the command was read: it needs the dx of
        // the command BEFORE)
        if (preops & kResPreAccept) { const int t = iy; iy = it; it = t; }     // swap(mBuffer, y), LS:1136
        int len = 1;
        if (action == kResEval || action == kResEvalSpec) {
            // (2) f(point) on the slice's rows -> trial residual, sum of squares (LS:1113-1115; LS:953-955 at entry). The point's
            // constants (Model::prepare) came with the command: workgroup 0 prepared them once for everybody (only the entry
            // evaluation, which no command precedes, prepares them here)
            if (round == 1) {
                if (tid == 0) Model::prepare(Xl, Cl);
                __syncthreads();
            }
            // kResEvalSpec: the thread that evaluated row i also forms the row's Broyden scale (LS:1003-1005: mBuffer = y_old -
            // y_new, += J dx, *= -1 / dx.dx) from ITS row of J -- a private dot product, no cross-lane sum; u is kept for the
            // products below and for the commit
            const bool spec_u = action == kResEvalSpec;
            double ss = 0;
            for (int i = tid; i < R; i += kResThreads) {
                const double v = i < nrows ? Model::eval(Dl + (size_t)i * ND, Cl) : 0.0;
                Yb[it * R + i] = v;
                ss = fma(v, v, ss);
                if (spec_u) {
                    const double* Ji = Jl + (size_t)i * JS;
                    double d0 = 0, d1 = 0;
#pragma unroll
                    for (int c = 0; c < N; c += 2) { d0 = fma(Ji[c], DXl[c], d0); if (c + 1 < N) d1 = fma(Ji[c + 1], DXl[c + 1], d1); }
                    const double t = (Yb[iy * R + i] - v) + (d0 + d1);
                    Yb[iu * R + i] = -inv_dd * t;
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) s_w[wave] = ss;
            __syncthreads();                                       // (also: every row of the trial residual is in LDS for step 3)
            if (tid == 0) res_st(a.partial + (size_t)wg * PL::STRIDE, wave_total(s_w));
            if (clk) { const long long t = wall_clock64(); t_w_eval += t - tw0; }
        } else if (action == kResFd && use_g) {
            // (2'') g(x, J), LS:1010-1014: one row per thread from the model's own derivative
            if constexpr (HAS_JAC) {
                if (tid == 0) Model::prepare(Xl, Cl);
                __syncthreads();
                for (int i = tid; i < nrows; i += kResThreads) Model::jac(Dl + (size_t)i * ND, Cl, Jl + (size_t)i * JS);
                __syncthreads();
            }
            if (clk) { const long long t = wall_clock64(); t_w_fd += t - tw0; }
        } else if (action == kResFd) {
            // (2') central differences at the command's point, LS:1018-1049: thread p prepares point p (2j: x + h e_j, 2j + 1:
            // x - h e_j, clipped to the bounds), then one row per thread
            if (tid < 2 * N) {
                const int j = tid >> 1;
                double p[N];
#pragma unroll
                for (int k = 0; k < N; ++k) p[k] = Xl[k];
                const double save = Xl[j];
                const double xmh = fmax(save - a.set.jacobianEpsilon, a.lower[j]);
                const double xph = fmin(save + a.set.jacobianEpsilon, a.upper[j]);
                const double twh = xph - xmh;
#pragma unroll
                for (int k = 0; k < N; ++k) p[k] = (k == j) ? ((tid & 1) ? xmh : xph) : p[k];
                Model::prepare(p, Cl + (size_t)tid * NCN);
                if ((tid & 1) == 0) INVl[j] = twh != 0 ? 1.0 / twh : 0.0;      // a collapsed interval: zero column, LS:1046
            }
            __syncthreads();
            // one (row, column) item at a time, consecutive threads on consecutive rows of one column (a thread that took whole
            // rows left a third of the workgroup idle in its last sweep: 391 rows on 256 threads)
            for (int e = tid; e < nrows * N; e += kResThreads) {
                const int j = e / nrows, i = e - j * nrows;
                const double* row = Dl + (size_t)i * ND;
                const double fp = Model::eval(row, Cl + (size_t)(2 * j) * NCN);
                const double fm = Model::eval(row, Cl + (size_t)(2 * j + 1) * NCN);
                const double inv = INVl[j];
                double v = fp;                                                 // copy, axpy(-1), scal(1 / twh): LS:1041-1047
                v += -1.0 * fm;
                Jl[(size_t)i * JS + j] = inv != 0 ? v * inv : 0.0;
            }
            __syncthreads();
            if (clk) { const long long t = wall_clock64(); t_w_fd += t - tw0; }
        }
        long long tp0 = 0;
        if (clk) tp0 = wall_clock64();
        if (action == kResEvalSpec || action == kResFd) {
            // (3) products of the slice on the matrix cores. Lane (q, p) holds J[4 s + q][16 c + p]: the A and the B operand of
            // v_mfma_f64_16x16x4 at once. kResEvalSpec: the rows are FIRST updated as the Broyden pass after an acceptance of
            // this trial would (LS:1006: J' = J + u dx^T with the row scales u of step 2) -- in registers only -- and the
            // products are those of J' with the trial residual.
            len = PL::LEN;
            const bool spec = action == kResEvalSpec;
            const int q = lane >> 4, p = lane & 15;
            const double* yv = Yb + (size_t)(spec ? it : iy) * R;
            double dxv[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) dxv[c] = DXl[16 * c + p];
            Acc acc[NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) acc[b] = Acc{0, 0, 0, 0};
            double jy[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) jy[c] = 0;
            // two k-steps per iteration (rows i0, i1): two independent chains of LDS reads, the 16-lane dot product and the MFMAs --
            // one chain alone leaves the wave waiting on its own latencies (R / 4 is a multiple of 2 x the waves)
            auto kstep = [&](const double (&vin)[NCB], double yi, double u) {
                double v[NCB];
#pragma unroll
                for (int c = 0; c < NCB; ++c) v[c] = vin[c];
                if (spec) {
#pragma unroll
                    for (int c = 0; c < NCB; ++c) v[c] = fma(u, dxv[c], v[c]);        // J' = J + u dx^T, LS:1006
                }
#pragma unroll
                for (int c1 = 0; c1 < NCB; ++c1)
#pragma unroll
                    for (int c2 = 0; c2 <= c1; ++c2) acc[c1 * (c1 + 1) / 2 + c2] = Mma<double>::mma(v[c1], v[c2], acc[c1 * (c1 + 1) / 2 + c2]);
#pragma unroll
                for (int c = 0; c < NCB; ++c) jy[c] = fma(v[c], yi, jy[c]);
            };
            for (int s = wave; s < R / 4; s += 2 * kResWaves) {
                const int i0 = 4 * s + q, i1 = i0 + 4 * kResWaves;
                double v0[NCB], v1[NCB];
#pragma unroll
                for (int c = 0; c < NCB; ++c) { v0[c] = Jl[(size_t)i0 * JS + 16 * c + p]; v1[c] = Jl[(size_t)i1 * JS + 16 * c + p]; }
                const double y0 = yv[i0], y1 = yv[i1], u0 = Yb[iu * R + i0], u1 = Yb[iu * R + i1];
                kstep(v0, y0, u0);
                kstep(v1, y1, u1);
            }
            // the four waves' accumulators through LDS, summed in a fixed order
            double* mine = RED + (size_t)wave * REDW;
#pragma unroll
            for (int b = 0; b < NBT; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) mine[b * 256 + Mma<double>::row(lane, r4) * 16 + p] = acc[b][r4];
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                double t = jy[c];
                t += wave_shfl_xor(t, 16);
                t += wave_shfl_xor(t, 32);
                if (q == 0) mine[NBT * 256 + 16 * c + p] = t;
            }
            __syncthreads();
            if (clk) t_w_mma += wall_clock64() - tp0;
            double* out = a.partial + (size_t)wg * PL::STRIDE;
            auto four = [&](int e) {                               // the waves' values in a fixed order
                double t = (RED[e] + RED[REDW + e]) + (RED[2 * REDW + e] + RED[3 * REDW + e]);
                if constexpr (kResWaves == 8)
                    t += (RED[4 * REDW + e] + RED[5 * REDW + e]) + (RED[6 * REDW + e] + RED[7 * REDW + e]);
                return t;
            };
            {
                int b = 0;
#pragma unroll
                for (int I = 0; I < NCB; ++I)
#pragma unroll
                    for (int J = 0; J <= I; ++J, ++b) {
                        const int r = tid >> 4, c = tid & 15;
                        if (tid < 256) {
                            const double v = four(b * 256 + tid);
                            if (I == J) { if (c <= r) res_st(out + PL::blk_base(b) + r * (r + 1) / 2 + c, v); }
                            else res_st(out + PL::blk_base(b) + tid, v);
                        }
                    }
            }
            if (tid < NC) res_st(out + PL::JY + tid, four(NBT * 256 + tid));
            if (action == kResFd && tid == 0) res_st(out, 0.0);
        }
        // (4) signal the group's leader
        res_drain();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.cnt + 32 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (clk) { const long long t = wall_clock64(); t_worker += t - tw0; t_w_prod += t - tp0; tw0 = t; }

        // =================================================================================== group leaders
        if (leader) {
            if (wave == 0) {
                const bool good = res_wait_ge(a.cnt + 32 * grp, (uint32_t)members * round, a.abort, spin_s);
                if (lane == 0) s_ok = good ? 1 : 0;
            }
            __syncthreads();
            if (!s_ok) { fail_out(1); return; }
            for (int e = tid; e < len; e += kResThreads) {
                double v[kResGroupMax];
#pragma unroll
                for (int j = 0; j < kResGroupMax; ++j) {
                    const int w = grp + NG * (j < members ? j : members - 1);
                    v[j] = res_ld(a.partial + (size_t)w * PL::STRIDE + e);
                }
                double s = v[0];
#pragma unroll
                for (int j = 1; j < kResGroupMax; ++j) s = j < members ? s + v[j] : s;
                res_st(a.gtotal + (size_t)grp * PL::STRIDE + e, s);
            }
            res_drain();
            __syncthreads();
            if (tid == 0) res_st(a.flag + 32 * grp, round);
        }
        if (clk) { const long long t = wall_clock64(); t_group += t - tw0; tw0 = t; }

        // =================================================================================== look-ahead (everybody but workgroup 0)
        // A rejected trial is followed by the SAME system with a larger damping, and the one-wave solve has already produced the
        // next levels of that ladder (solve_wave16.h). Their trial points came with the command: evaluate their sums of squares
        // now -- the leaders and workgroup 0 are at work for ~10 us, this workgroup would wait for the next command -- so that
        // workgroup 0 can book a rejected level WITHOUT a round of its own (LS:1112-1130 need nothing but the number).
        if constexpr (LOOK) {
            if (nlook > 0 && wg != 0) {
                look_sums(1, (int)nlook);
                if (tid < (int)nlook) res_st(a.look + (size_t)wg * 4 + tid, s_look[kResGroups + tid]);
                res_drain();
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(a.lcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        // =================================================================================== workgroup 0: the solver
        if (wg == 0) {
            if (wave == 0) {
                bool good = true;
                if (lane < NG) good = res_wait_ge(a.flag + 32 * lane, round, a.abort, spin_s);
                good = __all(good);
                if (lane == 0) s_ok = good ? 1 : 0;
            }
            __syncthreads();
            if (!s_ok) { fail_out(2); return; }
            if (clk) { const long long t = wall_clock64(); t_total_wait += t - tw0; tw0 = t; }
            double* TOT = RED;
            for (int e = tid; e < len; e += kResThreads) {
                double v[kResGroups];
#pragma unroll
                for (int g = 0; g < kResGroups; ++g) v[g] = res_ld(a.gtotal + (size_t)(g < NG ? g : NG - 1) * PL::STRIDE + e);
                double s = v[0];
#pragma unroll
                for (int g = 1; g < kResGroups; ++g) s = g < NG ? s + v[g] : s;
                TOT[e] = s;
            }
            __syncthreads();
            const double ss_total = TOT[0];
            if (len > 1) {
                // unpack into the pair that is NOT current: J^T J full symmetric n x n, J^T y
                double* JJn = JJp[cur ^ 1];
                double* Jyn = Jyp[cur ^ 1];
                int b = 0;
#pragma unroll
                for (int I = 0; I < NCB; ++I)
#pragma unroll
                    for (int J = 0; J <= I; ++J, ++b) {
                        const int r = tid >> 4, c = tid & 15;
                        const int gi = 16 * I + r, gj = 16 * J + c;
                        if (tid < 256 && gi < N && gj < N) {
                            double v;
                            if (I == J) v = c <= r ? TOT[PL::blk_base(b) + r * (r + 1) / 2 + c] : TOT[PL::blk_base(b) + c * (c + 1) / 2 + r];
                            else v = TOT[PL::blk_base(b) + tid];
                            JJn[(size_t)gi * JLD + gj] = v;
                            if (I != J) JJn[(size_t)gj * JLD + gi] = v;
                        }
                    }
                if (tid < N) Jyn[tid] = TOT[PL::JY + tid];
            }
            __syncthreads();

            if (clk) t_unpack += wall_clock64() - tw0;
            // ---- the loop of LS:972-1175, resumed where the last command left it
            uint32_t next_action = 0, next_pre = 0;
            enum { kTop, kAfterJac, kSolve, kAfterTrial, kCond, kDone } where;
            bool newJac = false;
            if (phase == 0) {                                                  // LS:953-971
                ++fCalls;
                residual = ss_total;
                fConverged = residual <= a.set.maxGoodResidual;
                needJac = true; age = maxAge; lambda = 0; mu = 1; iterations = 0; dx_dot = 0;
                next_pre |= kResPreAccept;                                     // the vector just evaluated IS y
                // validation of x0 and the bounds, LS:930-932 (the settings were checked on the host)
                bool finite = true, inb = true;
                for (int j = 0; j < N; ++j) {
                    const double xj = a.x[j];
                    if (!(-Lim<double>::inf() < xj && xj < Lim<double>::inf())) finite = false;
                    if (!(a.lower[j] <= xj) || !(xj <= a.upper[j])) inb = false;
                }
                if (tid < N) xsp[tid] = a.x[tid];
                __syncthreads();
                if (!finite) { status = mir_ls_badGuess; where = kDone; }
                else if (!inb) { status = mir_ls_badBounds; where = kDone; }
                else where = kTop;
            } else if (phase == 1) where = kAfterJac;
            else where = kAfterTrial;

            while (where != kDone && next_action == 0) {
                if (where == kCond) {                                          // LS:1175
                    if (iterations < a.maxIterations) where = kTop; else { where = kDone; break; }
                }
                if (where == kTop) {
                    ++n_passes;
                    if (fConverged) { status = mir_ls_fConverged; where = kDone; break; }                           // LS:974
                    if (!(lambda <= a.set.maxLambda)) { status = mir_ls_furtherImprovement; where = kDone; break; } // LS:979
                    if (mu > 16.0 && age) { needJac = true; age = maxAge; mu = 1; }                                 // LS:984
                    if (x_nan) { status = mir_ls_numericError; where = kDone; break; }                              // LS:990
                    newJac = false;
                    if (needJac) {                                             // LS:996
                        needJac = false;
                        if (age < maxAge) {                                    // Broyden, LS:999-1007: the speculative step becomes real
                            ++age;
                            next_pre |= kResPreCommitJ;
                            cur ^= 1;
                            null_lambda_for = 0; lad_valid = false;
                            ++n_br;
                            newJac = true;
                            trace(1, iterations, lambda, residual, 0, dx_dot);
                            where = kSolve;
                        } else {                                               // LS:1016-1050
                            age = 0;
                            next_action = kResFd;
                            phase = 1;
                            break;
                        }
                    } else where = kSolve;
                }
                if (where == kAfterJac) {
                    if (use_g) ++gCalls;                                       // LS:1013
                    else fCalls += N;                                          // LS:1049 (quirk Q5)
                    ++n_fd;
                    cur ^= 1;
                    null_lambda_for = 0; lad_valid = false;
                    newJac = true;
                    trace(0, iterations, lambda, residual, 0, dx_dot);
                    where = kSolve;
                }
                if (where == kSolve) {
                    // A pass after a rejection solves (J^T J + lambda I) dx = -J^T y again with a larger lambda. The solution of
                    // the (bound-constrained) QP satisfies |dx| <= 2 |J^T y| / lambda (its objective is <= 0, J^T J is positive
                    // semidefinite), so once 8 |J^T y|_2 / lambda < 2^-54 min |x_i| every component of the ROUNDED step
                    // (dx + x) - x, LS:1096-1097, is exactly zero: the pass is a null step whatever the solve returns -- trial == x,
                    // f(trial) is the residual we hold (pure callbacks, LS:73-80), improvement 0, rejected (LS:1125). Such passes
                    // -- most of the ~45 that end a noisy fit, quirk Q3 -- are booked without solving.
                    if (!newJac && null_lambda_for != 2) {
                        if (tid < kWave) {
                            double j2 = 0, xm = Lim<double>::inf(), dg = 0;
                            for (int j = lane; j < N; j += kWave) {
                                const double v = Jyp[cur][j];
                                j2 = fma(v, v, j2);
                                xm = fmin(xm, fabs(xsp[j]));
                                dg += fabs(JJp[cur][(size_t)j * JLD + j]);
                            }
                            j2 = wave_sum(j2);
                            dg = wave_sum(dg);
                            xm = -wave_max(-xm);
                            if (lane == 0) { s_w[0] = j2; s_w[1] = xm; s_w[2] = dg; }
                        }
                        __syncthreads();
                        const double j2 = s_w[0], xm = s_w[1], dg = s_w[2];
                        __syncthreads();
                        // (and lambda > 1e-6 trace(J^T J): P is then positive definite whatever rounding did to J^T J)
                        null_lambda = (j2 <= Lim<double>::max && dg <= Lim<double>::max && xm > 0)
                            ? fmax(8.0 * sqrt(j2) / (0x1p-54 * xm), 1e-6 * dg) : Lim<double>::inf();
                        null_lambda_for = 2;
                    }
                    if (!newJac && lambda > null_lambda && lambda >= a.set.minLambda && !(a.variant & kResVariantNoNullSkip)) {
                        ++fCalls;                                              // LS:1112
                        ++n_elided;
                        ++n_rej;
                        trace(2, iterations, lambda, residual, residual, 0.0);
                        lambda *= a.set.lambdaIncrease * mu;
                        mu *= 2;
                        where = kCond;
                        continue;
                    }
                    // LS:1052-1110, 1141-1142 on this workgroup: gradient test, lambda_0, solveBoxQP, rounding, trial, prediction
                    long long ts0 = 0;
                    if (clk) ts0 = wall_clock64();
                    const bool from_state = !(lambda >= a.set.minLambda);
                    ChainRec<double> rec{};
                    if constexpr (WAVE) {
                        // a pass after a rejection finds its step in the ladder the last solve made (same J^T J, J^T y, x; its
                        // damping bit for bit the ladder's next value): the steps are those of the one-by-one loop
                        const bool reuse = !newJac && lad_valid && lad_level + 1 < 4 && lrec[8 * (lad_level + 1)] == lambda
                            && lreci[4 * (lad_level + 1) + 3] != 0 && !from_state;
                        if (reuse) ++lad_level;
                        else {
                            if (wave == 0)
                                wave16_lm_solve_lds<N, BOUNDED>(JJp[cur], Jyp[cur], xsp, lop, upp, lambda, mu, newJac ? 1 : 0, from_state ? 1 : 0,
                                                                &a.set, dxp, trp, lrec, lreci);
                            __syncthreads();
                            lad_level = 0;
                            lad_valid = true;
                            look_n = 0;
                        }
                        rec.lambda = lrec[8 * lad_level]; rec.new_dx_dot = lrec[8 * lad_level + 1]; rec.predicted = lrec[8 * lad_level + 2];
                        rec.trial_xnorm = lrec[8 * lad_level + 3];
                        rec.qp_status = lreci[4 * lad_level]; rec.qp_iterations = lreci[4 * lad_level + 1]; rec.flags = lreci[4 * lad_level + 2];
                    } else {
                        if (tid == 0) a.st->lambda = lambda;
                        __syncthreads();
                        LmSolveArgs<double> sa{};
                        sa.JJ = a.JJ[cur]; sa.Jy = a.Jy[cur]; sa.x = a.xs; sa.lower = a.lower; sa.upper = a.upper;
                        sa.dx = a.dx; sa.trial = a.trial; sa.st = a.st; sa.rec = a.rec; sa.set = a.set; sa.n = N;
                        sa.sc[0] = a.sc; sa.lam[0] = lambda; sa.f_in_lds = 1;
                        sa.check_grad = newJac ? 1 : 0;
                        sa.lambda_from_state = from_state ? 1 : 0;
                        sa.lambda_from_device = 0;
                        lm_solve_body<double, NB, BOUNDED>(sa, 0, solve_smem);
                        __syncthreads();
                        rec = *a.rec;
                    }
                    if (clk) t_solve_body += wall_clock64() - ts0;
                    if (rec.flags & kFlagGradSmall) {                          // LS:1053-1062
                        if (age == 0) { status = mir_ls_gConverged; where = kDone; break; }
                        age = maxAge;
                        where = kCond;
                        continue;
                    }
                    lambda = rec.lambda;
                    if (rec.qp_status != 0 || (rec.flags & kFlagDxNaN)) { status = mir_ls_numericError; where = kDone; break; }   // LS:1080-1092
                    if (rec.qp_iterations > 0) ++n_qp;
                    s_ndd = rec.new_dx_dot; s_pred = rec.predicted; s_xnorm = rec.trial_xnorm; s_lam_used = lambda;
                    if (rec.flags & kFlagStepTooLong) {                        // LS:1101-1106
                        trace(4, iterations, lambda, residual, 0, s_ndd);
                        ++n_guard;
                        lambda *= a.set.lambdaIncrease * mu;
                        mu *= 2;
                        where = kCond;
                        continue;
                    }
                    ++fCalls;                                                  // LS:1112
                    if ((rec.flags & kFlagNullStep) && !(a.variant & kResVariantNoNullSkip)) {
                        // trial == x bit for bit and f is pure (LS:73-80): ||f(trial)||^2 is the residual we hold, improvement
                        // is 0, the pass is rejected (LS:1125) -- no round needed
                        ++n_elided;
                        ++n_rej;
                        trace(2, iterations, lambda, residual, residual, s_ndd);
                        lambda *= a.set.lambdaIncrease * mu;
                        mu *= 2;
                        where = kCond;
                        continue;
                    }
                    if constexpr (LOOK) {
                        if (look_n > 0 && lad_valid && lad_level > look_base && lad_level <= look_base + look_n) {
                            // this level's trial point went out with the last command: its sum of squares is there (or about
                            // to be). Own rows now; the others' in the order the leaders and the totals use (members of a group
                            // by rank, groups by number), so the number is the one a round would have produced.
                            long long tl0 = 0;
                            if (clk) tl0 = wall_clock64();
                            const int k = lad_level - look_base;               // 1 ... look_n
                            look_sums(k, 1);
                            if (!look_fetched) {
                                if (wave == 0) {
                                    const bool good = res_wait_ge(a.lcnt, look_target, a.abort, spin_s);
                                    if (lane == 0) s_ok = good ? 1 : 0;
                                }
                                __syncthreads();
                                if (!s_ok) { fail_out(4); return; }
                                look_fetched = true;
                            }
                            if (tid < NG) {
                                const int mem = (G - tid + NG - 1) / NG;
                                double v[kResGroupMax];
#pragma unroll
                                for (int j = 0; j < kResGroupMax; ++j) {
                                    const int w = tid + NG * (j < mem ? j : mem - 1);
                                    v[j] = w == 0 ? s_look[kResGroups] : res_ld(a.look + (size_t)w * 4 + (k - 1));
                                }
                                double sg = v[0];
#pragma unroll
                                for (int j = 1; j < kResGroupMax; ++j) sg = j < mem ? sg + v[j] : sg;
                                s_look[tid] = sg;
                            }
                            __syncthreads();
                            if (tid == 0) {
                                double tt = s_look[0];
                                for (int g = 1; g < NG; ++g) tt += s_look[g];
                                s_look[kResGroups + 3] = tt;
                            }
                            __syncthreads();
                            const double ahead = s_look[kResGroups + 3];
                            __syncthreads();
                            if (clk) t_look += wall_clock64() - tl0;
                            if (ahead <= Lim<double>::inf() && !(residual - ahead > 0)) {      // LS:1117, 1125-1130
                                trace(2, iterations, s_lam_used, residual, ahead, s_ndd);
                                ++n_rej;
                                ++n_look;
                                lambda *= a.set.lambdaIncrease * mu;
                                mu *= 2;
                                where = kCond;
                                continue;
                            }
                            // an improvement (or not a number): the round below evaluates the level again, with the products an
                            // accepted step needs
                        }
                    }
                    x_nan = (rec.flags & kFlagXNaN) != 0;                      // becomes x if accepted
                    next_action = age < maxAge ? kResEvalSpec : kResEval;
                    phase = 2;
                    break;
                }
                if (where == kAfterTrial) {
                    const double trialResidual = ss_total;
                    if (!(trialResidual <= Lim<double>::inf())) { status = mir_ls_numericError; where = kDone; break; }   // LS:1117
                    const double improvement = residual - trialResidual;
                    if (!(improvement > 0)) {                                  // LS:1125-1130
                        trace(2, iterations, s_lam_used, residual, trialResidual, s_ndd);
                        ++n_rej;
                        x_nan = false;
                        lambda *= a.set.lambdaIncrease * mu;
                        mu *= 2;
                        where = kCond;
                        continue;
                    }
                    needJac = true;                                            // LS:1132-1139
                    mu = 1;
                    ++iterations;
                    ++n_acc;
                    null_lambda_for = 0; lad_valid = false;
                    if (tid < N) xsp[tid] = trp[(WAVE ? 16 * lad_level : 0) + tid];
                    __syncthreads();
                    next_pre |= kResPreAccept;
                    residual = trialResidual;
                    fConverged = residual <= a.set.maxGoodResidual;
                    dx_dot = s_ndd;
                    trace(3, iterations, s_lam_used, residual, trialResidual, s_ndd);
                    if (!(s_pred > 0)) { status = mir_ls_furtherImprovement; where = kDone; break; }             // LS:1144-1148
                    const double rho = s_pred / improvement;                   // LS:1150 (quirk Q2)
                    if (rho < a.set.minStepQuality) { lambda *= a.set.lambdaIncrease * mu; mu *= 2; }
                    else if (rho >= a.set.goodStepQuality) lambda = fmax(a.set.lambdaDecrease * lambda * mu, a.set.minLambda);
                    const double dxn = sqrt(dx_dot);                           // LS:1164-1173 (quirk Q6)
                    if (!(dxn > a.set.absTolerance && s_xnorm > dxn * a.set.relTolerance)) {
                        if (age == 0) { status = mir_ls_xConverged; where = kDone; break; }
                        age = maxAge;
                    }
                    where = kCond;
                    continue;
                }
            }
            if (next_action == 0) next_action = kResExit;

            // ---- publish the command
            long long tpub0 = 0;
            if (clk) tpub0 = wall_clock64();
            int nl = 0;                                                        // look-ahead levels sent along
            if (next_action == kResExit) {
                if (tid < N) a.x[tid] = xsp[tid];
                if (tid == 0) {
                    mir_least_squares_result_d r;
                    r.status = (mir_least_squares_status)status; r.iterations = iterations; r.fCalls = fCalls; r.gCalls = gCalls;
                    r.residual = residual; r.lambda = lambda;
                    if (status == mir_ls_badGuess || status == mir_ls_badBounds) {           // LS:132-142: nothing was computed
                        r.iterations = 0; r.fCalls = 0; r.gCalls = 0; r.residual = Lim<double>::inf(); r.lambda = 0;
                    }
                    *a.result = r;
                    if (a.trace_count) *a.trace_count = tr_count;
                }
            } else {
                // further ladder levels to evaluate along (plain ones only: offered, solved, an ordinary step)
                if constexpr (LOOK) {
                    if (next_action != kResFd && lad_valid && !(a.variant & kResVariantNoLookahead)) {
                        for (int l = lad_level + 1; l < 4 && nl < kResLookMax; ++l) {
                            if (lreci[4 * l + 3] == 0 || lreci[4 * l] != 0) break;
                            if (lreci[4 * l + 2] & (kFlagDxNaN | kFlagStepTooLong | kFlagNullStep | kFlagGradSmall | kFlagXNaN)) break;
                            if (!(lrec[8 * l] <= a.set.maxLambda)) break;
                            ++nl;
                        }
                    }
                }
                look_n = nl; look_base = lad_level; look_fetched = false;
                if (nl > 0) look_target += (uint32_t)(G - 1);
                const double* pt = next_action == kResFd ? xsp : trp + (WAVE ? 16 * lad_level : 0);
                if (tid < kResNMax) {
                    res_st(a.cmd + tid, (unsigned long long)__double_as_longlong(tid < N ? pt[tid] : 0.0));
                    res_st(a.cmd + kResNMax + tid, (unsigned long long)__double_as_longlong(tid < N ? dxp[(WAVE ? 16 * lad_level : 0) + tid] : 0.0));
                }
                if (tid == 0) res_st(a.cmd + 2 * kResNMax, (unsigned long long)__double_as_longlong(1.0 / s_ndd));   // LS:1002
                if (next_action != kResFd) {
                    if (tid == 0) Model::prepare(pt, Cl);
                    if constexpr (LOOK) {
                        if (tid >= 1 && tid <= nl) Model::prepare(trp + 16 * (lad_level + tid), Cl + (size_t)tid * NCN);
                    }
                    __syncthreads();
                    if (tid < NCN) res_st(a.cmd + 2 * kResNMax + 2 + tid, (unsigned long long)__double_as_longlong(Cl[tid]));
                    if constexpr (LOOK) {
                        if (tid < nl * NCN) res_st(a.cmd + kResCmdWords + tid, (unsigned long long)__double_as_longlong(Cl[NCN + tid]));
                    }
                }
            }
            if (tid == 0)
                res_st(a.cmd + 2 * kResNMax + 1, (unsigned long long)(next_action | ((uint32_t)nl << 16)) | ((unsigned long long)next_pre << 32));
            res_drain();
            __syncthreads();
            if (tid == 0) res_st(a.seq, round);
            ++n_rounds;
            if (clk) { const long long t = wall_clock64(); t_solver += t - tw0; t_publish += t - tpub0; tw0 = t; }
        }

        // =================================================================================== everybody: the next command
        if (wave == 0) {
            const bool good = res_wait_ge(a.seq, round, a.abort, spin_s);
            if (lane == 0) s_ok = good ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) { fail_out(3); return; }
        if (tid < kResCmdWords) s_cmd[tid] = res_ld(a.cmd + tid);
        unsigned long long lw = 0;                                             // a look-ahead constant (used if the command says so)
        if constexpr (LOOK) {
            if (tid < kResLookMax * NCN) lw = res_ld(a.cmd + kResCmdWords + tid);
        }
        __syncthreads();
        action = (uint32_t)(s_cmd[2 * kResNMax + 1] & 0xffffull);
        nlook = LOOK ? (uint32_t)((s_cmd[2 * kResNMax + 1] >> 16) & 3ull) : 0u;
        preops = (uint32_t)(s_cmd[2 * kResNMax + 1] >> 32);
        if (clk) { const long long t = wall_clock64(); t_cmd_wait += t - tw0; }
        if (action == kResExit) break;
        inv_dd = __longlong_as_double((long long)s_cmd[2 * kResNMax]);
        // the commit of the speculative Broyden step, J += u dx^T (LS:1006): the same fma that formed J' for the products, with the
        // dx of the command that made the step -- so BEFORE DXl is replaced
        if (preops & kResPreCommitJ) {
            for (int e = tid; e < R * NC; e += kResThreads) {
                const int i = e / NC, c = e % NC;
                Jl[(size_t)i * JS + c] = fma(Yb[iu * R + i], DXl[c], Jl[(size_t)i * JS + c]);
            }
            preops &= ~kResPreCommitJ;
            __syncthreads();
        }
        if (tid < NC) {
            Xl[tid] = tid < N ? __longlong_as_double((long long)s_cmd[tid]) : 0.0;
            DXl[tid] = tid < N ? __longlong_as_double((long long)s_cmd[kResNMax + tid]) : 0.0;
        }
        if (action != kResFd && tid < NCN) Cl[tid] = __longlong_as_double((long long)s_cmd[2 * kResNMax + 2 + tid]);
        if constexpr (LOOK) {
            if (tid < (int)nlook * NCN) Cl[NCN + tid] = __longlong_as_double((long long)lw);
        }
        __syncthreads();
    }

    if (stamper && a.stats) {
        ResidentStats s{};
        s.rounds = n_rounds; s.passes = n_passes; s.accepted = n_acc; s.rejected = n_rej; s.step_guard_rejects = n_guard;
        s.jacobian_full = n_fd; s.jacobian_broyden = n_br; s.qp_active_set_passes = n_qp; s.elided_evaluations = n_elided;
        s.t_total = (uint64_t)(wall_clock64() - tk0); s.t_stage = (uint64_t)t_stage; s.t_worker = (uint64_t)t_worker;
        s.t_w_eval = (uint64_t)t_w_eval; s.t_w_fd = (uint64_t)t_w_fd; s.t_w_prod = (uint64_t)t_w_prod; s.t_w_mma = (uint64_t)t_w_mma;
        s.t_group = (uint64_t)t_group; s.t_total_wait = (uint64_t)t_total_wait; s.t_solver = (uint64_t)t_solver;
        s.t_solve_body = (uint64_t)t_solve_body; s.t_cmd_wait = (uint64_t)t_cmd_wait;
        s.lookahead_rejections = n_look; s.t_look = (uint64_t)t_look; s.t_unpack = (uint64_t)t_unpack; s.t_publish = (uint64_t)t_publish;
        s.abort_code = 0; s.grid = (uint32_t)G; s.rows = (uint32_t)a.rows; s.groups = (uint32_t)NG;
        *a.stats = s;
    }
}

}  // namespace mirlsq
