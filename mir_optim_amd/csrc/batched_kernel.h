// batched_kernel.h -- many small independent LM fits, ONE WAVEFRONT PER PROBLEM (BASELINE cfg 5:
// 4096 x (m = 512, n = 8), fp32). The whole loop of optimizeLeastSquaresImplGeneric!T
// (/root/reference/source/mir/optim/least_squares.d:877-1176) runs inside the kernel:
//   * a wave (= a workgroup) owns one problem; its J (m x n), y and the trial residual live in LDS, lane l owns rows
//     l, l + 64, ...; x, dx and all scalars are replicated in registers, J^T J and J^T y are held one row per lane;
//   * the residual model is a compile-time functor (no callback across the FFI in this entry);
//   * finite-difference Jacobian (LS:1018-1049), Broyden (LS:1002-1006), J^T J / J^T y as per-lane partial sums
//     + one wave reduction, the n x n damped solve with one matrix row per lane (?posvx semantics: equilibrate,
//     Cholesky, refine: posvx_rows), acceptance and the lambda/mu schedule exactly as the reference;
//   * all control flow is wave-uniform, there is no barrier and no host round trip.
// Problems whose step hits a finite bound are not finished here (BOXCQP's active-set loop is not part of this
// kernel): they return status kBatchedNeedsGeneral and the host entry re-solves them with the general solver.
#pragma once

#include "common.h"
#include "solve_types.h"

namespace mirlsq {

constexpr int kBatchedNeedsGeneral = -100;
constexpr int kBatchedNMax = 8;

enum : int { kModelExpDecay = 0, kModelExp3Affine = 1, kModelExpDecayPad8 = 2 };

// ---- residual models: r_i = Model::eval(t_i, basis_i, x) - data_i. The contract of a model (the built-in ones below and
// any user model handed to launch_batched<Model>, include/mir_optim_amd_batched.hpp) -- the compile-time counterpart of the
// reference's residual callback f(x, y) (least_squares.d:73-80), restricted to residuals that are a function of ONE abscissa:
//     static constexpr int n;      number of parameters, 1 <= n <= 8
//     static constexpr int nb;     per-row BASIS values that do not depend on the parameters (0 = none)
//     __device__ static void basis(float t, float* b);                       fills b[0 .. nb)
//     __device__ static float eval(float t, const float* b, const float* x); the model value at t; x has 8 entries, x[n..] = 0
// The basis values of every row are tabulated once per launch (k_batched_basis: rows x nb floats) and eval() reads the row's
// values instead of evaluating them again at every trial point and finite-difference point: the same floats enter the same
// expression. eval must be pure (the reference declares its callbacks pure) and wave-uniform in control flow.
struct ModelExpDecay {          // p0 exp(-t p1) + p2            (n = 3; reference unittest T5's family)
    static constexpr int n = 3, nb = 0;
    __device__ static inline void basis(float, float*) {}
    __device__ static inline float eval(float t, const float*, const float* x) { return x[0] * __expf(-t * x[1]) + x[2]; }
};
struct ModelExp3Affine {        // sum_{k<3} p_{2k} exp(-t p_{2k+1}) + p6 + p7 t   (n = 8)
    static constexpr int n = 8, nb = 0;
    __device__ static inline void basis(float, float*) {}
    __device__ static inline float eval(float t, const float*, const float* x)
    {
        return x[0] * __expf(-t * x[1]) + x[2] * __expf(-t * x[3]) + x[4] * __expf(-t * x[5]) + x[6] + x[7] * t;
    }
};
// exp(y) in float, the SAME bits on the device and on a host: every operation is written out (Cody-Waite reduction with a
// two-part ln 2, degree-7 Taylor polynomial on |r| <= ln 2 / 2: truncation 5e-9, ldexp) and is exactly rounded on both sides
// (fmaf, rintf, ldexpf); ~1.5 ulp. The float oracle's fused variant (oracle/lm_batched_fused.c) repeats it instruction for
// instruction, which libm's / the device library's expf would not allow (both are "<= 1 ulp", not the same ulp).
__host__ __device__ inline float det_expf(float y)
{
#pragma clang fp contract(off)
    // the ends behave as expf's do (round-4 advice: a clamp made exp(NaN) finite, so the reference's numericError exit on a
    // non-finite trial residual, LS:1117-1122, could not fire through this term): NaN stays NaN, overflow is +inf above
    // ln(FLT_MAX), and below -87 -- where the result would leave the normal range, which device and host ldexpf need not treat
    // alike -- the value is 0 (the true one is < 1.7e-38)
    if (!(y == y)) return y;
    if (y > 88.7228394f) return __builtin_huge_valf();
    if (y < -87.0f) return 0.0f;
    const float k = rintf(y * 1.44269504f);
    float r = __builtin_fmaf(k, -0.693145752f, y);
    r = __builtin_fmaf(k, -1.42860677e-06f, r);
    float p = 1.0f / 5040.0f;
    p = __builtin_fmaf(p, r, 1.0f / 720.0f);
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return ldexpf(p, (int)k);
}

// BASELINE cfg 5's well-conditioned n = 8 family ("p0 exp(-t p1) + p2 + 5-term variants padded to n = 8"): the
// exponential decay plus five terms that are LINEAR in their parameters (a two-frequency trigonometric pair and a slope),
// so the only nonlinearity is the decay and J^T J stays well conditioned in fp32. The four trigonometric values of a row are
// its basis (tabulated once a launch: device sinf / cosf; the fused oracle takes the table as an input). eval is ONE chain of
// fused multiply-adds around det_expf, nothing left to the compiler: the fused float oracle reproduces a fit of this model
// bit for bit (tests/test_gpu_batched.py); the libm oracle evaluates the same expression with expf and agrees to fp32 rounding.
struct ModelExpDecayPad8 {
    static constexpr int n = 8, nb = 4;
    __device__ static inline void basis(float t, float* b)
    {
        b[0] = sinf(2.0f * t); b[1] = cosf(2.0f * t); b[2] = sinf(5.0f * t); b[3] = cosf(5.0f * t);
    }
    __device__ static inline float eval(float t, const float* b, const float* x)
    {
#pragma clang fp contract(off)
        const float e = det_expf(-t * x[1]);
        float v = __builtin_fmaf(x[0], e, x[2]);
        v = __builtin_fmaf(x[3], b[0], v);
        v = __builtin_fmaf(x[4], b[1], v);
        v = __builtin_fmaf(x[5], b[2], v);
        v = __builtin_fmaf(x[6], b[3], v);
        return __builtin_fmaf(x[7], t, v);
    }
};
// the compiled-in models of mir_optimize_least_squares_batched_s by their MIR_LSQ_MODEL_* id
template <int ID> struct BuiltinModel;
template <> struct BuiltinModel<kModelExpDecay> { using type = ModelExpDecay; };
template <> struct BuiltinModel<kModelExp3Affine> { using type = ModelExp3Affine; };
template <> struct BuiltinModel<kModelExpDecayPad8> { using type = ModelExpDecayPad8; };

struct BatchedResult { int32_t status; uint32_t iterations, fCalls, gCalls; float residual, lambda; };

struct BatchedArgs {
    LmSettingsDev<float> set;
    uint32_t maxIterations, maxAge;
    int count, m;
    const float* t;        // m (shared) or count x m
    int t_stride;          // 0 = shared
    const float* data;     // count x m
    float* x;              // count x n, in/out
    const float* lower;    // n (shared)
    const float* upper;    // n
    BatchedResult* results;
    const float* basis;    // (t_stride ? count : 1) x m x nb: the model's per-row basis (k_batched_basis), nullptr when nb == 0
    uint64_t* timing;      // profiling builds (MIRLSQ_BATCHED_TIMING): 10 x count cycle counters (mir_lsq_batched_options.timing), else unused
    uint32_t variant;      // kBatchedNoLadder: one damping value per solve (A/B and the test of the ladder against it)
};
constexpr uint32_t kBatchedNoLadder = 1u;
constexpr uint32_t kBatchedAnalytic = 2u;      // MIR_LSQ_BATCHED_ANALYTIC_JACOBIAN: Model::grad instead of finite differences

// a model MAY provide the derivative of its value with respect to the parameters -- the reference's optional g callback
// (least_squares.d:80, 1010-1014):   __device__ static void grad(float t, const float* b, const float* x, float* gi /* n */);
template <class Model, class = void> struct batched_has_grad : std::false_type {};
template <class Model>
struct batched_has_grad<Model, std::void_t<decltype(Model::grad(0.0f, (const float*)nullptr, (const float*)nullptr, (float*)nullptr))>> : std::true_type {};

// one row of the basis table: 16-byte loads when the model has four values
template <int NB> struct BasisRow {
    float v[NB > 0 ? NB : 1];
    __device__ inline void load(const float* table, int i)
    {
        if constexpr (NB == 4) {
            const float4 q = reinterpret_cast<const float4*>(table)[i];
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) v[k] = table[(size_t)i * NB + k];
        }
    }
};

template <class Model>
__global__ __launch_bounds__(256) void k_batched_basis(const float* __restrict__ t, float* __restrict__ table, size_t rows)
{
    constexpr int NB = Model::nb;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
        float b[NB > 0 ? NB : 1];
        Model::basis(t[i], b);
#pragma unroll
        for (int k = 0; k < NB; ++k) table[i * NB + k] = b[k];
    }
}

// ---- the damped n x n solve with ONE ROW PER LANE ------------------------------------------------------------------------
// ?posvx('E','L') as the oracle restates it (oracle/lm_oracle_impl.inc, lmo_posvx: ?poequ, ?laqsy, ?potf2, ?potrs, ?porfs;
// the condition estimate is left out: the reference accepts info = n + 1, boxcqp.d:212, and reads no other output of it).
// Lane r = lane & 7 of every group of eight lanes holds row r of each matrix (eight registers a matrix instead of the 36 of a
// private copy per lane, which took 256 + 29 registers and ~2000 instructions a solve) and component r of each vector; the
// eight groups of a wave compute the same thing. A value of another row comes through v_readlane (lanes 0..7 hold every row).
// Every element sees the operations of the oracle's loops in the oracle's order: the left-looking sums of ?potf2
// (`s -= F[i][k] F[j][k]`, k ascending) are applied one k at a time to the whole trailing part; the forward sweep of ?potrs
// runs by columns, its backward sweep (a chain that can only start when z[i + 1] is known) in lane i on column i of the
// factor. Every multiply-add is ONE fused operation (__builtin_fmaf), division and square root are the IEEE ones: the result
// equals, bit for bit, the oracle's float ?posvx written with fmaf in the same loops (oracle/lm_oracle.c, lmo_posvx_fused_s;
// tests/test_gpu_batched.py) -- and this file does not depend on which products the compiler chooses to fuse.
// Divisions and square roots per solve: 44 + 9 sequences (a copy per lane: 76 + 16).
__device__ inline float lane_get(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
// a[r], r = lane & 7, as a chain of selects on VALUES: taking the array by reference lets the optimiser turn the chain into one
// load at a computed address, which pins the whole array in scratch memory
__device__ inline float row_pick8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, int r)
{
    float v = a0;
    v = (r == 1) ? a1 : v; v = (r == 2) ? a2 : v; v = (r == 3) ? a3 : v; v = (r == 4) ? a4 : v;
    v = (r == 5) ? a5 : v; v = (r == 6) ? a6 : v; v = (r == 7) ? a7 : v;
    return v;
}
#define MIRLSQ_ROW_PICK(a, r) row_pick8((a)[0], (a)[1], (a)[2], (a)[3], (a)[4], (a)[5], (a)[6], (a)[7], (r))
// max / min over the eight rows (every lane gets it): the values repeat with period 8 along a 16-lane DPP row, so the
// rotations by 4, 2, 1 pair each lane with the rows r ^ 4, then r ^ 2, r ^ 1. fmaxf / fminf: a NaN operand is ignored.
__device__ inline float rows_max(float v)
{
    v = fmaxf(v, dpp_row_ror<4>(v)); v = fmaxf(v, dpp_row_ror<2>(v)); v = fmaxf(v, dpp_row_ror<1>(v));
    return v;
}
__device__ inline float rows_min(float v)
{
    v = fminf(v, dpp_row_ror<4>(v)); v = fminf(v, dpp_row_ror<2>(v)); v = fminf(v, dpp_row_ror<1>(v));
    return v;
}

// ?posvx('E','L'), n = N <= NMAX = 8, FOUR SYSTEMS A WAVE: each row of 16 lanes (a DPP row; r = lane & 7, the upper eight lanes
// repeat the lower eight) solves its own system, so that one call serves a ladder of four damping values (k_lm_batched).
// Prow: the full symmetric row r of this group's P; rhs_r: component r of its right-hand side. x: the group's solution in every
// lane of the group. Returns the group's info in every lane of the group. There is no branch on a group's data: a group whose
// factorization fails keeps computing on values nobody reads.
template <int N, int NMAX>
__device__ inline int posvx_rows(const float (&Prow)[NMAX], float rhs_r, int r, float (&x)[NMAX])
{
    static_assert(NMAX == 8, "row r = lane & 7");
    const float eps = Lim<float>::eps / 2, safmin = Lim<float>::min_normal;
    const bool live = r < N;
    const float d_r = MIRLSQ_ROW_PICK(Prow, r);
    // ?poequ
    const float smin = rows_min(live ? d_r : Lim<float>::inf());
    const float amax = rows_max(live ? d_r : -Lim<float>::inf());
    const bool pos = smin > 0;
    const float scond = sqrtf(smin) / sqrtf(amax);
    const float s_r = (pos && live) ? 1.0f / sqrtf(d_r) : 1.0f;
    const float small = safmin / Lim<float>::eps, large = 1.0f / small;
    const bool rcequ = pos && !(scond >= 0.1f && amax >= small && amax <= large);
    // ?laqsy
    float Arow[NMAX], Frow[NMAX], Fcol[NMAX], s[NMAX];           // Fcol[k] = F[k][r], k > r: column r of the factor, for L^T
    static_for<NMAX>([&](auto K) {
        constexpr int k = K.value;
        s[k] = dpp_row_bcast<k>(s_r);
        const float v = Prow[k];
        Arow[k] = (live && k < N) ? (rcequ ? s[k] * s_r * v : v) : (r == k ? 1.0f : 0.0f);
        Frow[k] = Arow[k];
        Fcol[k] = 0.0f;
    });
    const float b_r = live ? (rcequ ? s_r * rhs_r : rhs_r) : 0.0f;
    // ?potf2 'L': after step j, Frow[jj] (jj > j) of row r >= jj holds A[r][jj] - sum_{k <= j} F[r][k] F[jj][k]
    int info = 0;
    static_for<NMAX>([&](auto J) {
        constexpr int j = J.value;
        if constexpr (j < N) {
            float ajj = dpp_row_bcast<j>(Frow[j]);
            info = (info == 0 && !(ajj > 0)) ? j + 1 : info;
            ajj = sqrtf(ajj);
            const float q = Frow[j] / ajj;
            Frow[j] = (r == j) ? ajj : q;                          // rows above the diagonal carry values nobody reads
            static_for<NMAX>([&](auto JJ) {
                constexpr int jj = JJ.value;
                if constexpr (jj > j && jj < N) {
                    const float ljj = dpp_row_bcast<jj>(Frow[j]);  // F[jj][j]
                    Frow[jj] = __builtin_fmaf(-Frow[j], ljj, Frow[jj]);
                    Fcol[jj] = (r == j) ? ljj : Fcol[jj];
                }
            });
        }
    });
    const float fd_r = MIRLSQ_ROW_PICK(Frow, r);                  // F[r][r]
    // ?potrs. L y = v by columns: row r takes `t -= F[r][i] y[i]` at step i (ascending i, as in the oracle's row loop).
    // L^T z = y: row i needs t = y[i] - sum_{k > i} F[k][i] z[k] with k ascending, a chain that can only start when z[i + 1]
    // is known: lane i runs it on its column of the factor and the group's z[k].
    auto potrs = [&](float v_r, float (&z)[NMAX]) {
        static_for<NMAX>([&](auto I) {
            constexpr int i = I.value;
            if constexpr (i < N) {
                const float yi = dpp_row_bcast<i>(v_r / fd_r);
                const float upd = __builtin_fmaf(-Frow[i], yi, v_r);
                v_r = (r == i) ? yi : (r > i ? upd : v_r);
            }
        });
#pragma unroll
        for (int i = 0; i < NMAX; ++i) z[i] = 0.0f;
        static_for<NMAX>([&](auto II) {
            constexpr int i = NMAX - 1 - II.value;
            if constexpr (i < N) {
                float t = v_r;
#pragma unroll
                for (int k = 0; k < NMAX; ++k) if (k > i && k < N) t = __builtin_fmaf(-Fcol[k], z[k], t);
                z[i] = dpp_row_bcast<i>(t / fd_r);
            }
        });
    };
    potrs(b_r, x);
    // ?porfs: the loop runs while any group refines; a group that has stopped keeps its solution
    const float safe1 = (float)(N + 1) * safmin, safe2 = safe1 / eps;
    float lstres = 3;
    bool active = true;
    for (int count = 1;; ++count) {
        float ri = b_r, wi = fabsf(b_r);
#pragma unroll
        for (int k = 0; k < NMAX; ++k) if (k < N) {
            ri = __builtin_fmaf(-Arow[k], x[k], ri);
            wi = __builtin_fmaf(fabsf(Arow[k]), fabsf(x[k]), wi);
        }
        const bool big = wi > safe2;
        const float q = (big ? fabsf(ri) : fabsf(ri) + safe1) / (big ? wi : wi + safe1);
        const float berr = rows_max(live ? q : 0.0f);
        active = active && berr > eps && 2 * berr <= lstres && count <= 5;
        if (__builtin_amdgcn_ballot_w64(active) == 0) break;
        float c[NMAX];
        potrs(live ? ri : 0.0f, c);
#pragma unroll
        for (int i = 0; i < NMAX; ++i) x[i] = active ? x[i] + c[i] : x[i];
        lstres = active ? berr : lstres;
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i) x[i] = rcequ ? s[i] * x[i] : x[i];
    return info;
}

// unit-test entry of posvx_rows: four systems a wave; P count x 64 (row-major, lower triangle read), rhs and x count x 8
template <int N>
__global__ __launch_bounds__(64) void k_posvx_rows(const float* __restrict__ P, const float* __restrict__ rhs, int count,
                                                   float* __restrict__ x, int* __restrict__ info)
{
    const int lane = threadIdx.x, r = lane & 7, g = lane >> 4;
    for (int p0 = 4 * blockIdx.x; p0 < count; p0 += 4 * gridDim.x) {
        const int p = p0 + g < count ? p0 + g : count - 1;            // a short last wave repeats the last system
        float Prow[8], sol[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) Prow[k] = (r < N && k < N) ? P[(size_t)p * 64 + (k <= r ? r * 8 + k : k * 8 + r)] : 0.0f;
        const int rc = posvx_rows<N, 8>(Prow, r < N ? rhs[(size_t)p * 8 + r] : 0.0f, r, sol);
        if ((lane & 15) == 0 && p0 + g < count) {
            info[p] = rc;
#pragma unroll
            for (int k = 0; k < 8; ++k) x[(size_t)p * 8 + k] = (rc == 0 && k < N) ? sol[k] : 0.0f;
        }
    }
}

// -DMIRLSQ_BATCHED_TIMING: per problem, the shader-clock cycles (s_memtime) spent in [0] residual evaluations, [1] Jacobian
// refreshes (FD or Broyden), [2] J^T J / J^T y with its reductions, [3] damped solves, [4] the whole fit, and [5] the number of
// solve calls, [6] a trial's preparation, [7] an accepted step's bookkeeping, written to BatchedArgs::timing (10 x uint64 a problem). A profiling build only (scripts/probes/cfg5_phases.py).
#ifdef MIRLSQ_BATCHED_TIMING
#define MIRLSQ_T0() const uint64_t t0_ = __builtin_readcyclecounter()
#define MIRLSQ_T1(k) tacc[k] += __builtin_readcyclecounter() - t0_
#else
#define MIRLSQ_T0() ((void)0)
#define MIRLSQ_T1(k) ((void)0)
#endif

template <class Model>
__global__ __launch_bounds__(64, Model::n <= 4 ? 4 : 2) void k_lm_batched(BatchedArgs a)   // waves per SIMD the LDS slices allow at m = 512
{
    // Nothing in this body is left to the compiler's choice of what to fuse: contraction is off and every multiply-add that is
    // meant to be ONE rounding is a __builtin_fmaf. The arithmetic of a fit is then a fixed sequence of IEEE operations that
    // oracle/lm_batched_fused.c repeats on the host (per-lane partial sums, the butterfly of wave_sum): bit-identical results.
#pragma clang fp contract(off)
    constexpr int N = Model::n;
    constexpr int NMAX = kBatchedNMax;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    // one problem per (single-wave) workgroup: the dispatcher hands a finished wave's slot to the next problem, so a
    // long fit delays nobody (four problems per workgroup held three slots until the slowest of the four was done)
    const int lane = threadIdx.x;
    const int prob = blockIdx.x;
    const int m = a.m;
    float* Jl = reinterpret_cast<float*>(smem_b);                          // J: m x N row-major
    float* yv = Jl + (size_t)N * m;
    float* mB = yv + m;
    const float* tp = a.t + (size_t)(a.t_stride ? prob : 0) * a.t_stride;
    const float* dp = a.data + (size_t)prob * m;
    constexpr int NB = Model::nb;
    const float* bp = NB ? a.basis + (size_t)(a.t_stride ? prob : 0) * a.t_stride * NB : nullptr;
    const LmSettingsDev<float>& S = a.set;

    float x[NMAX], lo[NMAX], up[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        x[j] = j < N ? a.x[(size_t)prob * N + j] : 0.0f;
        lo[j] = j < N ? a.lower[j] : -Lim<float>::inf();
        up[j] = j < N ? a.upper[j] : Lim<float>::inf();
    }
#ifdef MIRLSQ_BATCHED_TIMING
    uint64_t tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t tstart = __builtin_readcyclecounter();
#endif
    BatchedResult ret;
    ret.status = -26;   // numericError, LS:132
    ret.iterations = 0; ret.fCalls = 0; ret.gCalls = 0;
    ret.residual = Lim<float>::inf(); ret.lambda = 0;

    // The lane's rows (lane, lane + 64, ...) are taken in chunks of UNR: the loads of a chunk are issued together (index clamped
    // to the last row: always a valid address), then the rows are used in order, so a wave does not pay one memory latency a
    // row. The sum of squares is accumulated in the order of the plain loop.
    auto feval = [&](const float (&p)[NMAX], float* dst) -> float {      // dst = f(p); returns ||f||^2
        constexpr int UNR = 8;
        float ss = 0;
        for (int base = lane; base - lane < m; base += kWave * UNR) {
            float tv[UNR], dv[UNR], rv[UNR];
            BasisRow<NB> bv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = min(base + kWave * u, m - 1);
                tv[u] = tp[i]; dv[u] = dp[i];
                bv[u].load(bp, i);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) rv[u] = Model::eval(tv[u], bv[u].v, p) - dv[u];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = base + kWave * u;
                if (i < m) dst[i] = rv[u];
                ss = i < m ? __builtin_fmaf(rv[u], rv[u], ss) : ss;
            }
        }
        return wave_sum(ss);
    };

    // validation LS:930-943 (settings were checked on the host; x / bounds here)
    bool finite = true, inb = true;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) if (j < N) {
        if (!(-Lim<float>::inf() < x[j] && x[j] < Lim<float>::inf())) finite = false;
        if (!(lo[j] <= x[j]) || !(x[j] <= up[j])) inb = false;
    }
    if (m == 0 || !finite) ret.status = -31;           // badGuess
    else if (!inb) ret.status = -32;                   // badBounds
    else {
        constexpr bool HAS_GRAD = batched_has_grad<Model>::value;
        const bool use_g = HAS_GRAD && (a.variant & kBatchedAnalytic) != 0;  // g of LS:1010-1014 (the launcher refuses it without grad)
        const uint32_t maxAge = a.maxAge ? a.maxAge : (use_g ? 3u : 2u * N);     // LS:945
        { MIRLSQ_T0(); ret.residual = feval(x, yv); MIRLSQ_T1(0); }       // LS:953-955
        ++ret.fCalls;
        bool fConverged = ret.residual <= S.maxGoodResidual;
        bool needJacobian = true;
        uint32_t age = maxAge;
        // J^T J and J^T y live one ROW per lane (row r = lane & 7 in every group of eight lanes), as posvx_rows wants them
        const int r = lane & 7;
        float dx[NMAX], JJrow[NMAX], Jy_r = 0;
#pragma unroll
        for (int j = 0; j < NMAX; ++j) { dx[j] = 0; JJrow[j] = 0; }
        float lad_x[NMAX], lad_lam[4] = {0, 0, 0, 0};      // the ladder of solutions (group g of the wave: level g), see below
#pragma unroll
        for (int j = 0; j < NMAX; ++j) lad_x[j] = 0;
        int lad_info = 0, lad_level = 0;
        bool lad_valid = false;
        const int lad_depth = (a.variant & kBatchedNoLadder) ? 1 : 4;
        float dx_dot = 0, mu = 1, lambda = 0;
        ret.status = -1;                                                   // maxIterations, LS:971
        do {
            if (fConverged) { ret.status = 3; break; }                     // LS:974
            if (!(lambda <= S.maxLambda)) { ret.status = 0; break; }       // LS:979
            if (mu > 16.0f && age) { needJacobian = true; age = maxAge; mu = 1; }   // LS:984
            {
                bool nan = false;
#pragma unroll
                for (int j = 0; j < NMAX; ++j) if (j < N && !(x[j] <= x[j])) nan = true;
                if (nan) { ret.status = -26; break; }                      // LS:990
            }
            if (needJacobian) {                                            // LS:996
                needJacobian = false;
                MIRLSQ_T0();
                if (age < maxAge) {                                        // Broyden LS:999-1007
                    age++;
                    const float d = 1.0f / dx_dot;
                    for (int i = lane; i < m; i += kWave) {
                        float* Ji = Jl + (size_t)i * N;
                        float dot = 0;
#pragma unroll
                        for (int j = 0; j < N; ++j) dot = __builtin_fmaf(Ji[j], dx[j], dot);
                        const float t = (mB[i] - yv[i]) + dot;             // mB holds the previous residual
                        const float u = -d * t;
#pragma unroll
                        for (int j = 0; j < N; ++j) Ji[j] = __builtin_fmaf(u, dx[j], Ji[j]);
                    }
                } else if (use_g) {                                        // g(x, J), LS:1010-1014
                    age = 0;
                    if constexpr (HAS_GRAD) {
                        for (int i = lane; i < m; i += kWave) {
                            BasisRow<NB> b;
                            b.load(bp, i);
                            float gi[NMAX];
#pragma unroll
                            for (int j = 0; j < NMAX; ++j) gi[j] = 0;
                            Model::grad(tp[i], b.v, x, gi);
#pragma unroll
                            for (int j = 0; j < N; ++j) Jl[(size_t)i * N + j] = gi[j];
                        }
                    }
                    ++ret.gCalls;                                          // LS:1013
                } else {                                                   // FD LS:1016-1050
                    age = 0;
                    // the n central differences of a row share its t, d and basis: rows outside, columns inside
                    float xph[NMAX], xmh[NMAX], inv[NMAX];
#pragma unroll
                    for (int j = 0; j < NMAX; ++j) {
                        xmh[j] = fmaxf(x[j] - S.jacobianEpsilon, lo[j]);
                        xph[j] = fminf(x[j] + S.jacobianEpsilon, up[j]);
                        const float twh = xph[j] - xmh[j];
                        inv[j] = twh != 0 ? 1.0f / twh : 0.0f;             // a zero-width interval: the column is zero, LS:1045
                    }
                    for (int i = lane; i < m; i += kWave) {
                        BasisRow<NB> b;
                        b.load(bp, i);
                        const float ti = tp[i], di = dp[i];
                        float p[NMAX];
#pragma unroll
                        for (int k = 0; k < NMAX; ++k) p[k] = x[k];
#pragma unroll
                        for (int j = 0; j < N; ++j) {
                            p[j] = xph[j];
                            const float fp = Model::eval(ti, b.v, p) - di;
                            p[j] = xmh[j];
                            const float fm = Model::eval(ti, b.v, p) - di;
                            p[j] = x[j];
                            const float v = fp - fm;
                            Jl[(size_t)i * N + j] = inv[j] != 0 ? v * inv[j] : 0.0f;
                        }
                    }
                    ret.fCalls += N;                                       // LS:1049 (quirk Q5)
                }
#ifdef MIRLSQ_BATCHED_TIMING
                const uint64_t t1_ = __builtin_readcyclecounter();
                tacc[1] += t1_ - t0_;
#endif
                // Jy = J^T y (LS:1052) and JJ = J^T J lower (LS:1065) in one sweep over the lane's rows
                float accJ[NMAX][NMAX], accy[NMAX];
#pragma unroll
                for (int j = 0; j < NMAX; ++j) { accy[j] = 0;
#pragma unroll
                    for (int k = 0; k < NMAX; ++k) accJ[j][k] = 0; }
                for (int i = lane; i < m; i += kWave) {
                    const float* Ji = Jl + (size_t)i * N;
                    const float yi = yv[i];
                    float row[NMAX];
#pragma unroll
                    for (int j = 0; j < NMAX; ++j) row[j] = j < N ? Ji[j] : 0.0f;
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        accy[j] = __builtin_fmaf(row[j], yi, accy[j]);
#pragma unroll
                        for (int k = 0; k <= j; ++k) accJ[j][k] = __builtin_fmaf(row[j], row[k], accJ[j][k]);
                    }
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float ty = wave_sum(accy[j]);
                    Jy_r = (r == j) ? ty : Jy_r;
#pragma unroll
                    for (int k = 0; k <= j; ++k) {
                        const float t = wave_sum(accJ[j][k]);                  // element (j, k) and its mirror (k, j)
                        JJrow[k] = (r == j) ? t : JJrow[k];
                        if (k != j) JJrow[j] = (r == k) ? t : JJrow[j];
                    }
                }
                lad_valid = false;                                             // J^T J has changed
#ifdef MIRLSQ_BATCHED_TIMING
                tacc[2] += __builtin_readcyclecounter() - t1_;
#endif
                const float gmax = lane_get(rows_max(fabsf(Jy_r)), 0);         // rows >= N hold zeros
                if (!(gmax > S.gradTolerance)) {                           // LS:1053-1062
                    if (age == 0) { ret.status = 2; break; }
                    age = maxAge;
                    continue;
                }
            }
            if (!(lambda >= S.minLambda)) {                                // LS:1067-1072
                // the largest diagonal element (a sum of squares: its own absolute value; a NaN is skipped as by `>`)
                const float best = lane_get(rows_max(r < N ? fabsf(MIRLSQ_ROW_PICK(JJrow, r)) : -1.0f), 0);
                const float val = best < 0 ? 0.0f : best;
                lambda = 0.001f * val;
                if (!(lambda >= S.minLambda)) lambda = 1;
            }
            // LS:1079-1080 (-> QP:194). The four 16-lane groups of the wave solve with lambda and with the three values the
            // rejection rule (LS:1101-1106, 1125-1130: lambda *= lambdaIncrease mu, mu *= 2) would make of it next, at the
            // cost of one solve; a rejected step then finds its solution ready. A level is used only while J^T J is the one
            // the ladder was built on and lambda is bit for bit the ladder's value: the steps are those of the one-by-one loop.
            if (!(lad_valid && lad_level < lad_depth && lambda == lad_lam[lad_level])) {
                float l = lambda, mm = mu;
#pragma unroll
                for (int g = 0; g < 4; ++g) { lad_lam[g] = l; l *= S.lambdaIncrease * mm; mm *= 2; }
                const int g = lane >> 4;
                const float mine = g == 0 ? lad_lam[0] : (g == 1 ? lad_lam[1] : (g == 2 ? lad_lam[2] : lad_lam[3]));
                float Prow[NMAX];
#pragma unroll
                for (int k = 0; k < NMAX; ++k) Prow[k] = JJrow[k] + ((k == r && r < N) ? mine : 0.0f);   // (Q1)
                MIRLSQ_T0();
                lad_info = posvx_rows<N, NMAX>(Prow, -Jy_r, r, lad_x);
                MIRLSQ_T1(3);
#ifdef MIRLSQ_BATCHED_TIMING
                ++tacc[5];
#endif
                lad_level = 0;
                lad_valid = true;
            }
#ifdef MIRLSQ_BATCHED_TIMING
            const uint64_t t6_ = __builtin_readcyclecounter();
#endif
            float sol[NMAX];
            const int lad_lane = 16 * lad_level++;
            const int info = __builtin_amdgcn_readlane(lad_info, lad_lane);
#pragma unroll
            for (int j = 0; j < NMAX; ++j) sol[j] = lane_get(lad_x[j], lad_lane);
            if (info != 0) { ret.status = -26; break; }
            bool feasible = true, nan = false;
#pragma unroll
            for (int j = 0; j < NMAX; ++j) if (j < N) {
                if (!((lo[j] - x[j]) <= sol[j] && sol[j] <= (up[j] - x[j]))) feasible = false;   // QP:216-219
                if (!(sol[j] <= sol[j])) nan = true;
            }
            if (nan) { ret.status = -26; break; }                          // LS:1087
            if (!feasible) { ret.status = kBatchedNeedsGeneral; break; }   // active-set loop: general solver
            float trial[NMAX], ndd = 0;
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                float d = sol[j] + x[j];                                   // LS:1096-1097
                d = d - x[j];
                sol[j] = j < N ? d : 0.0f;
                ndd = __builtin_fmaf(sol[j], sol[j], ndd);
                trial[j] = fmaxf(fminf(sol[j] + x[j], up[j]), lo[j]);      // LS:1108-1110
            }
            if (!(sqrtf(ndd) < S.maxStep)) { lambda *= S.lambdaIncrease * mu; mu *= 2; continue; }   // LS:1101-1106
            ++ret.fCalls;                                                  // LS:1112-1115
#ifdef MIRLSQ_BATCHED_TIMING
            tacc[6] += __builtin_readcyclecounter() - t6_;
#endif
            // the trial residual goes to the buffer that is NOT the current y
            float trialResidual;
            { MIRLSQ_T0(); trialResidual = feval(trial, mB); MIRLSQ_T1(0); }
            if (!(trialResidual <= Lim<float>::inf())) { ret.status = -26; break; }   // LS:1117
            const float improvement = ret.residual - trialResidual;
#ifdef MIRLSQ_BATCHED_TIMING
            const uint64_t t7_ = __builtin_readcyclecounter();
#endif
            if (!(improvement > 0)) { lambda *= S.lambdaIncrease * mu; mu *= 2; continue; }   // LS:1125-1130
            needJacobian = true;                                           // LS:1132-1139
            mu = 1;
            ret.iterations++;
#pragma unroll
            for (int j = 0; j < NMAX; ++j) { x[j] = trial[j]; dx[j] = sol[j]; }
            { float* tmp = yv; yv = mB; mB = tmp; }                        // swap(mBuffer, y): mB = previous residual
            ret.residual = trialResidual;
            fConverged = ret.residual <= S.maxGoodResidual;
            dx_dot = ndd;
            float pred = 0;                                                // LS:1141-1142 (undamped JJ)
            {
                float tj = 0;                                              // row r of J^T J dx + 2 J^T y, then the dot with dx
#pragma unroll
                for (int k = 0; k < NMAX; ++k) tj = __builtin_fmaf(JJrow[k], dx[k], tj);
                tj = tj + 2 * Jy_r;
#pragma unroll
                for (int j = 0; j < NMAX; ++j) pred = __builtin_fmaf(lane_get(tj, j), dx[j], pred);
            }
            pred = -pred;
            if (!(pred > 0)) { ret.status = 0; break; }                    // LS:1144-1148
            const float rho = pred / improvement;                          // LS:1150 (Q2)
            if (rho < S.minStepQuality) { lambda *= S.lambdaIncrease * mu; mu *= 2; }
            else if (rho >= S.goodStepQuality) lambda = fmaxf(S.lambdaDecrease * lambda * mu, S.minLambda);
            float xn = 0;
#pragma unroll
            for (int j = 0; j < NMAX; ++j) xn = __builtin_fmaf(x[j], x[j], xn);
#ifdef MIRLSQ_BATCHED_TIMING
            tacc[7] += __builtin_readcyclecounter() - t7_;
#endif
            if (!(sqrtf(dx_dot) > S.absTolerance && sqrtf(xn) > sqrtf(dx_dot) * S.relTolerance)) {   // LS:1164-1173 (Q6)
                if (age == 0) { ret.status = 1; break; }
                age = maxAge;
                continue;
            }
        } while (ret.iterations < a.maxIterations);                        // LS:1175
        ret.lambda = lambda;
    }
#ifdef MIRLSQ_BATCHED_TIMING
    tacc[4] = __builtin_readcyclecounter() - tstart;
    if (lane == 0 && a.timing) for (int k = 0; k < 10; ++k) a.timing[(size_t)prob * 10 + k] = tacc[k];
#endif
    if (lane == 0) {
        a.results[prob] = ret;
#pragma unroll
        for (int j = 0; j < N; ++j) a.x[(size_t)prob * N + j] = x[j];
    }
}


// residual of one problem as a DEVICE callback body (used when a batched problem falls back to the general solver)
template <class Model>
__global__ __launch_bounds__(256) void k_batched_model_eval(const float* __restrict__ t, const float* __restrict__ d,
                                                            const float* __restrict__ x, float* __restrict__ y, int m)
{
    constexpr int NB = Model::nb;
    float p[kBatchedNMax];
#pragma unroll
    for (int j = 0; j < kBatchedNMax; ++j) p[j] = j < Model::n ? x[j] : 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        float b[NB > 0 ? NB : 1];
        Model::basis(t[i], b);
        y[i] = Model::eval(t[i], b, p) - d[i];
    }
}

}  // namespace mirlsq
