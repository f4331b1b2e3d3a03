// batched_kernel.h -- many small independent LM fits, ONE WAVEFRONT PER PROBLEM (BASELINE cfg 5:
// 4096 x (m = 512, n = 8), fp32). The whole loop of optimizeLeastSquaresImplGeneric!T
// (/root/reference/source/mir/optim/least_squares.d:877-1176) runs inside the kernel:
//   * a wave owns one problem; its J (m x n), y and the trial residual live in the wave's slice of LDS,
//     lane l owns rows l, l + 64, ...; x, dx, J^T J, J^T y and all scalars are replicated in registers;
//   * the residual model is a compile-time functor (no callback across the FFI in this entry);
//   * finite-difference Jacobian (LS:1018-1049), Broyden (LS:1002-1006), J^T J / J^T y as per-lane partial sums
//     + one wave reduction, the n x n damped solve redundantly in every lane (?posvx semantics: equilibrate,
//     Cholesky, refine), acceptance and the lambda/mu schedule exactly as the reference;
//   * all control flow is wave-uniform, there is no barrier and no host round trip.
// Problems whose step hits a finite bound are not finished here (BOXCQP's active-set loop is not part of this
// kernel): they return status kBatchedNeedsGeneral and the host entry re-solves them with the general solver.
#pragma once

#include "common.h"
#include "solve_kernel.h"

namespace mirlsq {

constexpr int kBatchedNeedsGeneral = -100;
constexpr int kBatchedNMax = 8;

enum : int { kModelExpDecay = 0, kModelExp3Affine = 1, kModelExpDecayPad8 = 2 };

// residual models: r = model(t, x) - d
template <int MODEL> struct BatchedModel;
template <> struct BatchedModel<kModelExpDecay> {      // p0 exp(-t p1) + p2            (n = 3; reference unittest T5's family)
    static constexpr int n = 3;
    __device__ static inline float eval(float t, const float* x) { return x[0] * __expf(-t * x[1]) + x[2]; }
};
template <> struct BatchedModel<kModelExp3Affine> {    // sum_{k<3} p_{2k} exp(-t p_{2k+1}) + p6 + p7 t   (n = 8)
    static constexpr int n = 8;
    __device__ static inline float eval(float t, const float* x)
    {
        return x[0] * __expf(-t * x[1]) + x[2] * __expf(-t * x[3]) + x[4] * __expf(-t * x[5]) + x[6] + x[7] * t;
    }
};

// BASELINE cfg 5's well-conditioned n = 8 family ("p0 exp(-t p1) + p2 + 5-term variants padded to n = 8"): the
// exponential decay plus five terms that are LINEAR in their parameters (a two-frequency trigonometric pair and a slope),
// so the only nonlinearity is the decay and J^T J stays well conditioned in fp32. Precise expf / sinf / cosf (the float
// oracle evaluates the same expression with libm).
template <> struct BatchedModel<kModelExpDecayPad8> {
    static constexpr int n = 8;
    __device__ static inline float eval(float t, const float* x)
    {
        return x[0] * expf(-t * x[1]) + x[2] + x[3] * sinf(2.0f * t) + x[4] * cosf(2.0f * t) + x[5] * sinf(5.0f * t)
             + x[6] * cosf(5.0f * t) + x[7] * t;
    }
};

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
};

// ?posvx('E','L') for n <= NMAX, redundantly in every lane. P: lower triangle meaningful. Returns info.
template <int NMAX>
__device__ inline int posvx_small(int n, const float (&P)[NMAX][NMAX], const float (&rhs)[NMAX], float (&x)[NMAX])
{
    const float eps = Lim<float>::eps / 2, safmin = Lim<float>::min_normal;
    float A[NMAX][NMAX], F[NMAX][NMAX], s[NMAX], b[NMAX];
    float smin = Lim<float>::inf(), amax = -Lim<float>::inf();
#pragma unroll
    for (int i = 0; i < NMAX; ++i) if (i < n) { smin = fminf(smin, P[i][i]); amax = fmaxf(amax, P[i][i]); }
    bool rcequ = false;
    if (smin > 0) {
        const float scond = sqrtf(smin) / sqrtf(amax);
#pragma unroll
        for (int i = 0; i < NMAX; ++i) s[i] = i < n ? 1.0f / sqrtf(P[i][i]) : 1.0f;
        const float small = safmin / Lim<float>::eps, large = 1.0f / small;
        rcequ = !(scond >= 0.1f && amax >= small && amax <= large);
    } else {
#pragma unroll
        for (int i = 0; i < NMAX; ++i) s[i] = 1.0f;
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i)
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            const float v = (j <= i) ? P[i][j] : P[j][i];
            A[i][j] = (i < n && j < n) ? (rcequ ? s[j] * s[i] * v : v) : (i == j ? 1.0f : 0.0f);
        }
#pragma unroll
    for (int i = 0; i < NMAX; ++i) b[i] = i < n ? (rcequ ? s[i] * rhs[i] : rhs[i]) : 0.0f;
    // ?potf2 'L'
#pragma unroll
    for (int i = 0; i < NMAX; ++i)
#pragma unroll
        for (int j = 0; j < NMAX; ++j) F[i][j] = A[i][j];
    int info = 0;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        if (info == 0 && j < n) {
            float ajj = F[j][j];
#pragma unroll
            for (int k = 0; k < NMAX; ++k) if (k < j) ajj -= F[j][k] * F[j][k];
            if (!(ajj > 0)) { info = j + 1; }
            else {
                ajj = sqrtf(ajj);
                F[j][j] = ajj;
#pragma unroll
                for (int i = 0; i < NMAX; ++i) if (i > j && i < n) {
                    float v = F[i][j];
#pragma unroll
                    for (int k = 0; k < NMAX; ++k) if (k < j) v -= F[i][k] * F[j][k];
                    F[i][j] = v / ajj;
                }
            }
        }
    }
    if (info != 0) return info;
    auto potrs = [&](float (&v)[NMAX]) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i) if (i < n) {
            float t = v[i];
#pragma unroll
            for (int k = 0; k < NMAX; ++k) if (k < i) t -= F[i][k] * v[k];
            v[i] = t / F[i][i];
        }
#pragma unroll
        for (int ii = 0; ii < NMAX; ++ii) {
            const int i = NMAX - 1 - ii;
            if (i < n) {
                float t = v[i];
#pragma unroll
                for (int k = 0; k < NMAX; ++k) if (k > i && k < n) t -= F[k][i] * v[k];
                v[i] = t / F[i][i];
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NMAX; ++i) x[i] = b[i];
    potrs(x);
    // ?porfs
    const float safe1 = (float)(n + 1) * safmin, safe2 = safe1 / eps;
    float lstres = 3;
    for (int count = 1;; ++count) {
        float r[NMAX], berr = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            float ri = b[i], wi = fabsf(b[i]);
#pragma unroll
            for (int k = 0; k < NMAX; ++k) if (k < n) { ri -= A[i][k] * x[k]; wi += fabsf(A[i][k]) * fabsf(x[k]); }
            r[i] = i < n ? ri : 0.0f;
            if (i < n) {
                const float q = (wi > safe2) ? fabsf(ri) / wi : (fabsf(ri) + safe1) / (wi + safe1);
                berr = fmaxf(berr, q);
            }
        }
        if (berr > eps && 2 * berr <= lstres && count <= 5) {
            potrs(r);
#pragma unroll
            for (int i = 0; i < NMAX; ++i) x[i] += r[i];
            lstres = berr;
            continue;
        }
        break;
    }
    if (rcequ) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i) x[i] = s[i] * x[i];
    }
    return 0;
}

template <int MODEL>
__global__ __launch_bounds__(256) void k_lm_batched(BatchedArgs a)
{
    constexpr int N = BatchedModel<MODEL>::n;
    constexpr int NMAX = kBatchedNMax;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int prob = blockIdx.x * 4 + wave;
    if (prob >= a.count) return;
    const int m = a.m;
    float* Jl = reinterpret_cast<float*>(smem_b) + (size_t)wave * (N + 2) * m;   // J: m x N row-major
    float* yv = Jl + (size_t)N * m;
    float* mB = yv + m;
    const float* tp = a.t + (size_t)(a.t_stride ? prob : 0) * a.t_stride;
    const float* dp = a.data + (size_t)prob * m;
    const LmSettingsDev<float>& S = a.set;

    float x[NMAX], lo[NMAX], up[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        x[j] = j < N ? a.x[(size_t)prob * N + j] : 0.0f;
        lo[j] = j < N ? a.lower[j] : -Lim<float>::inf();
        up[j] = j < N ? a.upper[j] : Lim<float>::inf();
    }
    BatchedResult ret;
    ret.status = -26;   // numericError, LS:132
    ret.iterations = 0; ret.fCalls = 0; ret.gCalls = 0;
    ret.residual = Lim<float>::inf(); ret.lambda = 0;

    auto feval = [&](const float (&p)[NMAX], float* dst) -> float {      // dst = f(p); returns ||f||^2
        float ss = 0;
        for (int i = lane; i < m; i += kWave) {
            const float r = BatchedModel<MODEL>::eval(tp[i], p) - dp[i];
            dst[i] = r;
            ss += r * r;
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
        const uint32_t maxAge = a.maxAge ? a.maxAge : 2 * N;               // LS:945 (no analytic Jacobian here)
        ret.residual = feval(x, yv);                                       // LS:953-955
        ++ret.fCalls;
        bool fConverged = ret.residual <= S.maxGoodResidual;
        bool needJacobian = true;
        uint32_t age = maxAge;
        float dx[NMAX], Jy[NMAX], JJ[NMAX][NMAX];
#pragma unroll
        for (int j = 0; j < NMAX; ++j) { dx[j] = 0; Jy[j] = 0; }
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
                if (age < maxAge) {                                        // Broyden LS:999-1007
                    age++;
                    const float d = 1.0f / dx_dot;
                    for (int i = lane; i < m; i += kWave) {
                        float* Ji = Jl + (size_t)i * N;
                        float dot = 0;
#pragma unroll
                        for (int j = 0; j < N; ++j) dot += Ji[j] * dx[j];
                        const float t = (mB[i] - yv[i]) + dot;             // mB holds the previous residual
                        const float u = -d * t;
#pragma unroll
                        for (int j = 0; j < N; ++j) Ji[j] += u * dx[j];
                    }
                } else {                                                   // FD LS:1016-1050
                    age = 0;
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        float p[NMAX];
#pragma unroll
                        for (int k = 0; k < NMAX; ++k) p[k] = x[k];
                        const float save = x[j];
                        float xmh = save - S.jacobianEpsilon, xph = save + S.jacobianEpsilon;
                        xmh = fmaxf(xmh, lo[j]);
                        xph = fminf(xph, up[j]);
                        const float twh = xph - xmh;
                        if (twh != 0) {
                            const float inv = 1.0f / twh;
                            for (int i = lane; i < m; i += kWave) {
                                p[j] = xph;
                                const float fp = BatchedModel<MODEL>::eval(tp[i], p) - dp[i];
                                p[j] = xmh;
                                const float fm = BatchedModel<MODEL>::eval(tp[i], p) - dp[i];
                                float v = fp;
                                v += -1.0f * fm;
                                Jl[(size_t)i * N + j] = v * inv;
                            }
                        } else {
                            for (int i = lane; i < m; i += kWave) Jl[(size_t)i * N + j] = 0.0f;
                        }
                    }
                    ret.fCalls += N;                                       // LS:1049 (quirk Q5)
                }
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
                        accy[j] += row[j] * yi;
#pragma unroll
                        for (int k = 0; k <= j; ++k) accJ[j][k] += row[j] * row[k];
                    }
                }
#pragma unroll
                for (int j = 0; j < NMAX; ++j) {
                    Jy[j] = j < N ? wave_sum(accy[j]) : 0.0f;
#pragma unroll
                    for (int k = 0; k < NMAX; ++k) JJ[j][k] = (j < N && k <= j) ? wave_sum(accJ[j][k]) : 0.0f;
                }
                float gmax = 0;
#pragma unroll
                for (int j = 0; j < NMAX; ++j) if (j < N) gmax = fmaxf(gmax, fabsf(Jy[j]));
                if (!(gmax > S.gradTolerance)) {                           // LS:1053-1062
                    if (age == 0) { ret.status = 2; break; }
                    age = maxAge;
                    continue;
                }
            }
            if (!(lambda >= S.minLambda)) {                                // LS:1067-1072
                float best = -1, val = 0;
#pragma unroll
                for (int j = 0; j < NMAX; ++j) if (j < N && fabsf(JJ[j][j]) > best) { best = fabsf(JJ[j][j]); val = JJ[j][j]; }
                lambda = 0.001f * val;
                if (!(lambda >= S.minLambda)) lambda = 1;
            }
            float P[NMAX][NMAX], rhs[NMAX], sol[NMAX];
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                rhs[j] = -Jy[j];
#pragma unroll
                for (int k = 0; k < NMAX; ++k) P[j][k] = JJ[j][k] + ((j == k && j < N) ? lambda : 0.0f);   // LS:1079 (Q1)
            }
            const int info = posvx_small<NMAX>(N, P, rhs, sol);            // LS:1080 -> QP:194
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
                ndd += sol[j] * sol[j];
                trial[j] = fmaxf(fminf(sol[j] + x[j], up[j]), lo[j]);      // LS:1108-1110
            }
            if (!(sqrtf(ndd) < S.maxStep)) { lambda *= S.lambdaIncrease * mu; mu *= 2; continue; }   // LS:1101-1106
            ++ret.fCalls;                                                  // LS:1112-1115
            // the trial residual goes to the buffer that is NOT the current y
            const float trialResidual = feval(trial, mB);
            if (!(trialResidual <= Lim<float>::inf())) { ret.status = -26; break; }   // LS:1117
            const float improvement = ret.residual - trialResidual;
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
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                float tj = 0;
#pragma unroll
                for (int k = 0; k < NMAX; ++k) tj += ((k <= j) ? JJ[j][k] : JJ[k][j]) * dx[k];
                tj = tj + 2 * Jy[j];
                pred += tj * dx[j];
            }
            pred = -pred;
            if (!(pred > 0)) { ret.status = 0; break; }                    // LS:1144-1148
            const float rho = pred / improvement;                          // LS:1150 (Q2)
            if (rho < S.minStepQuality) { lambda *= S.lambdaIncrease * mu; mu *= 2; }
            else if (rho >= S.goodStepQuality) lambda = fmaxf(S.lambdaDecrease * lambda * mu, S.minLambda);
            float xn = 0;
#pragma unroll
            for (int j = 0; j < NMAX; ++j) xn += x[j] * x[j];
            if (!(sqrtf(dx_dot) > S.absTolerance && sqrtf(xn) > sqrtf(dx_dot) * S.relTolerance)) {   // LS:1164-1173 (Q6)
                if (age == 0) { ret.status = 1; break; }
                age = maxAge;
                continue;
            }
        } while (ret.iterations < a.maxIterations);                        // LS:1175
        ret.lambda = lambda;
    }
    if (lane == 0) {
        a.results[prob] = ret;
#pragma unroll
        for (int j = 0; j < N; ++j) a.x[(size_t)prob * N + j] = x[j];
    }
}


// residual of one problem as a DEVICE callback body (used when a batched problem falls back to the general solver)
template <int MODEL>
__global__ __launch_bounds__(256) void k_batched_model_eval(const float* __restrict__ t, const float* __restrict__ d,
                                                            const float* __restrict__ x, float* __restrict__ y, int m)
{
    float p[kBatchedNMax];
#pragma unroll
    for (int j = 0; j < kBatchedNMax; ++j) p[j] = j < BatchedModel<MODEL>::n ? x[j] : 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        y[i] = BatchedModel<MODEL>::eval(t[i], p) - d[i];
}

}  // namespace mirlsq
