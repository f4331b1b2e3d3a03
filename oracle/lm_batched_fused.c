/* lm_batched_fused.c -- TEST INFRASTRUCTURE (never linked into the product): the float Levenberg-Marquardt fit of
 * /root/reference/source/mir/optim/least_squares.d:877-1176 written with the ARITHMETIC of the device's one-wavefront-per-problem
 * kernel (mir_optim_amd/csrc/batched_kernel.h, k_lm_batched), operation for operation:
 *   - every multiply-add the kernel fuses is an fmaf here, nothing else is fused (this file is compiled with -ffp-contract=off);
 *   - sums over the m rows are the kernel's: 64 per-lane partial sums over rows lane, lane + 64, ... in ascending order, then the
 *     butterfly of wave_sum (csrc/common.h: row rotations by 8, 4, 2, 1 inside each 16-lane row, rows 0 + 1 and 2 + 3, then both);
 *   - the damped solve is lmo_posvx_fused_s (lm_oracle.c), which the kernel's posvx_rows equals bit for bit;
 *   - the residual model is cfg 5's padded exponential decay with the kernel's det_expf (explicit Cody-Waite + Taylor, the same
 *     bits on both sides) and the row's four trigonometric basis values TAKEN AS INPUT (the device tabulates them once a launch).
 * The control flow is the reference's (the LS: line numbers are those of lm_oracle_impl.inc / the kernel); the lambda ladder of the
 * kernel (four damping values solved at once) is bitwise equivalent to solving them one by one and is not restated.
 * A fit that would need the active-set loop (a finite bound hit) returns LMO_BATCHED_NEEDS_GENERAL like the kernel. */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lm_oracle.h"

#define NPAR 8
#define LANES 64

static float det_expf(float y)                       /* batched_kernel.h: det_expf */
{
    if (!(y == y)) return y;                         /* NaN stays NaN, +inf above ln(FLT_MAX), 0 below -87: as the kernel */
    if (y > 88.7228394f) return HUGE_VALF;
    if (y < -87.0f) return 0.0f;
    const float k = rintf(y * 1.44269504f);
    float r = fmaf(k, -0.693145752f, y);
    r = fmaf(k, -1.42860677e-06f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return ldexpf(p, (int)k);
}

static float model_eval(float t, const float* b, const float* x)     /* ModelExpDecayPad8::eval */
{
    const float e = det_expf(-t * x[1]);
    float v = fmaf(x[0], e, x[2]);
    v = fmaf(x[3], b[0], v);
    v = fmaf(x[4], b[1], v);
    v = fmaf(x[5], b[2], v);
    v = fmaf(x[6], b[3], v);
    return fmaf(x[7], t, v);
}

/* common.h: sum16 then the two cross-row steps, read at lane 63. sum16 in lane l of a row: v += v[(l - 8) & 15], then 4, 2, 1
 * (rotations inside the row; addition is commutative, so only the pairing matters). */
static float wave_sum64(const float* v)
{
    float D[4];
    for (int row = 0; row < 4; ++row) {
        float a[16], b[16];
        for (int l = 0; l < 16; ++l) a[l] = v[16 * row + l];
        for (int sh = 8; sh >= 1; sh >>= 1) {
            for (int l = 0; l < 16; ++l) b[l] = a[l] + a[(l - sh) & 15];
            memcpy(a, b, sizeof a);
        }
        D[row] = a[15];
    }
    return (D[3] + D[2]) + (D[1] + D[0]);
}

typedef struct {
    int m;
    const float *t, *basis, *data;
} fit_ctx;

/* feval: dst = f(p), returns ||f||^2 as the kernel sums it */
static float feval(const fit_ctx* c, const float* p, float* dst)
{
    float ss[LANES];
    for (int l = 0; l < LANES; ++l) ss[l] = 0;
    for (int i = 0; i < c->m; ++i) {
        const float rv = model_eval(c->t[i], c->basis + 4 * (size_t)i, p) - c->data[i];
        dst[i] = rv;
        ss[i % LANES] = fmaf(rv, rv, ss[i % LANES]);
    }
    return wave_sum64(ss);
}

int lmo_optimize_batched_fused_pad8_s(const lmo_settings_s* S, int m, const float* t, const float* basis, const float* data,
                                      float* x_io, const float* lower, const float* upper, lmo_result_s* out)
{
    const int N = NPAR;
    fit_ctx c = {m, t, basis, data};
    lmo_result_s ret;
    ret.status = -26; ret.iterations = 0; ret.fCalls = 0; ret.gCalls = 0; ret.residual = INFINITY; ret.lambda = 0;
    float x[NPAR], lo[NPAR], up[NPAR];
    for (int j = 0; j < N; ++j) { x[j] = x_io[j]; lo[j] = lower[j]; up[j] = upper[j]; }
    float* J = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1) * N);
    float* ybuf = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1) * 2);
    if (!J || !ybuf) { free(J); free(ybuf); return -1; }
    float *yv = ybuf, *mB = ybuf + (m > 0 ? m : 1);

    int finite = 1, inb = 1;
    for (int j = 0; j < N; ++j) {
        if (!(-INFINITY < x[j] && x[j] < INFINITY)) finite = 0;
        if (!(lo[j] <= x[j]) || !(x[j] <= up[j])) inb = 0;
    }
    if (m == 0 || !finite) ret.status = -31;
    else if (!inb) ret.status = -32;
    else {
        const uint32_t maxAge = S->maxAge ? S->maxAge : 2 * N;                 /* LS:945 */
        ret.residual = feval(&c, x, yv);                                     /* LS:953-955 */
        ++ret.fCalls;
        int fConverged = ret.residual <= S->maxGoodResidual;
        int needJacobian = 1;
        uint32_t age = maxAge;
        float dx[NPAR], JJ[NPAR][NPAR], Jy[NPAR];
        memset(dx, 0, sizeof dx); memset(JJ, 0, sizeof JJ); memset(Jy, 0, sizeof Jy);
        float dx_dot = 0, mu = 1, lambda = 0;
        ret.status = -1;                                                     /* LS:971 */
        do {
            if (fConverged) { ret.status = 3; break; }                       /* LS:974 */
            if (!(lambda <= S->maxLambda)) { ret.status = 0; break; }        /* LS:979 */
            if (mu > 16.0f && age) { needJacobian = 1; age = maxAge; mu = 1; }   /* LS:984 */
            {
                int nan = 0;
                for (int j = 0; j < N; ++j) if (!(x[j] <= x[j])) nan = 1;
                if (nan) { ret.status = -26; break; }                        /* LS:990 */
            }
            if (needJacobian) {                                              /* LS:996 */
                needJacobian = 0;
                if (age < maxAge) {                                          /* Broyden LS:999-1007 */
                    age++;
                    const float d = 1.0f / dx_dot;
                    for (int i = 0; i < m; ++i) {
                        float* Ji = J + (size_t)i * N;
                        float dot = 0;
                        for (int j = 0; j < N; ++j) dot = fmaf(Ji[j], dx[j], dot);
                        const float tt = (mB[i] - yv[i]) + dot;
                        const float u = -d * tt;
                        for (int j = 0; j < N; ++j) Ji[j] = fmaf(u, dx[j], Ji[j]);
                    }
                } else {                                                     /* FD LS:1016-1050 */
                    age = 0;
                    float xph[NPAR], xmh[NPAR], inv[NPAR];
                    for (int j = 0; j < N; ++j) {
                        xmh[j] = fmaxf(x[j] - S->jacobianEpsilon, lo[j]);
                        xph[j] = fminf(x[j] + S->jacobianEpsilon, up[j]);
                        const float twh = xph[j] - xmh[j];
                        inv[j] = twh != 0 ? 1.0f / twh : 0.0f;
                    }
                    for (int i = 0; i < m; ++i) {
                        const float* b = basis + 4 * (size_t)i;
                        const float ti = t[i], di = data[i];
                        float p[NPAR];
                        memcpy(p, x, sizeof p);
                        for (int j = 0; j < N; ++j) {
                            p[j] = xph[j];
                            const float fp = model_eval(ti, b, p) - di;
                            p[j] = xmh[j];
                            const float fm = model_eval(ti, b, p) - di;
                            p[j] = x[j];
                            const float v = fp - fm;
                            J[(size_t)i * N + j] = inv[j] != 0 ? v * inv[j] : 0.0f;
                        }
                    }
                    ret.fCalls += N;                                         /* LS:1049 (quirk Q5) */
                }
                /* Jy = J^T y, JJ = J^T J: per-lane partial sums, then the butterfly */
                static float accy[NPAR][LANES], accJ[NPAR][NPAR][LANES];
                memset(accy, 0, sizeof accy); memset(accJ, 0, sizeof accJ);
                for (int i = 0; i < m; ++i) {
                    const float* Ji = J + (size_t)i * N;
                    const float yi = yv[i];
                    const int l = i % LANES;
                    for (int j = 0; j < N; ++j) {
                        accy[j][l] = fmaf(Ji[j], yi, accy[j][l]);
                        for (int k = 0; k <= j; ++k) accJ[j][k][l] = fmaf(Ji[j], Ji[k], accJ[j][k][l]);
                    }
                }
                for (int j = 0; j < N; ++j) {
                    Jy[j] = wave_sum64(accy[j]);
                    for (int k = 0; k <= j; ++k) { const float s = wave_sum64(accJ[j][k]); JJ[j][k] = s; JJ[k][j] = s; }
                }
                float gmax = 0;                                              /* rows_max of |Jy|: fmaxf, NaN ignored */
                for (int j = 0; j < N; ++j) gmax = fmaxf(gmax, fabsf(Jy[j]));
                if (!(gmax > S->gradTolerance)) {                            /* LS:1053-1062 */
                    if (age == 0) { ret.status = 2; break; }
                    age = maxAge;
                    continue;
                }
            }
            if (!(lambda >= S->minLambda)) {                                 /* LS:1067-1072 */
                float best = -1.0f;
                for (int j = 0; j < N; ++j) best = fmaxf(best, fabsf(JJ[j][j]));
                const float val = best < 0 ? 0.0f : best;
                lambda = 0.001f * val;
                if (!(lambda >= S->minLambda)) lambda = 1;
            }
            /* LS:1079-1080: P = JJ + lambda I, solve P d = -Jy (posvx_rows == lmo_posvx_fused_s) */
            float P[NPAR * NPAR], rhs[NPAR], sol[NPAR];
            for (int i = 0; i < N; ++i) {
                for (int k = 0; k < N; ++k) P[i + (size_t)k * N] = JJ[i][k] + (i == k ? lambda : 0.0f);
                rhs[i] = -Jy[i];
            }
            const int info = lmo_posvx_fused_s(N, P, N, rhs, sol, NULL);
            if (info != 0) { ret.status = -26; break; }
            int feasible = 1, nan = 0;
            for (int j = 0; j < N; ++j) {
                if (!((lo[j] - x[j]) <= sol[j] && sol[j] <= (up[j] - x[j]))) feasible = 0;      /* QP:216-219 */
                if (!(sol[j] <= sol[j])) nan = 1;
            }
            if (nan) { ret.status = -26; break; }                            /* LS:1087 */
            if (!feasible) { ret.status = LMO_BATCHED_NEEDS_GENERAL; break; }
            float trial[NPAR], ndd = 0;
            for (int j = 0; j < N; ++j) {
                volatile float d = sol[j] + x[j];                            /* LS:1096-1097 */
                d = d - x[j];
                sol[j] = d;
                ndd = fmaf(sol[j], sol[j], ndd);
                trial[j] = fmaxf(fminf(sol[j] + x[j], up[j]), lo[j]);        /* LS:1108-1110 */
            }
            if (!(sqrtf(ndd) < S->maxStep)) { lambda *= S->lambdaIncrease * mu; mu *= 2; continue; }   /* LS:1101-1106 */
            ++ret.fCalls;                                                    /* LS:1112-1115 */
            const float trialResidual = feval(&c, trial, mB);
            if (!(trialResidual <= INFINITY)) { ret.status = -26; break; }   /* LS:1117 */
            const float improvement = ret.residual - trialResidual;
            if (!(improvement > 0)) { lambda *= S->lambdaIncrease * mu; mu *= 2; continue; }           /* LS:1125-1130 */
            needJacobian = 1;                                                /* LS:1132-1139 */
            mu = 1;
            ret.iterations++;
            for (int j = 0; j < N; ++j) { x[j] = trial[j]; dx[j] = sol[j]; }
            { float* tmp = yv; yv = mB; mB = tmp; }
            ret.residual = trialResidual;
            fConverged = ret.residual <= S->maxGoodResidual;
            dx_dot = ndd;
            float pred = 0;                                                  /* LS:1141-1142 */
            for (int j = 0; j < N; ++j) {
                float tj = 0;
                for (int k = 0; k < N; ++k) tj = fmaf(JJ[j][k], dx[k], tj);
                tj = tj + 2 * Jy[j];
                pred = fmaf(tj, dx[j], pred);
            }
            pred = -pred;
            if (!(pred > 0)) { ret.status = 0; break; }                      /* LS:1144-1148 */
            const float rho = pred / improvement;                            /* LS:1150 (Q2) */
            if (rho < S->minStepQuality) { lambda *= S->lambdaIncrease * mu; mu *= 2; }
            else if (rho >= S->goodStepQuality) lambda = fmaxf(S->lambdaDecrease * lambda * mu, S->minLambda);
            float xn = 0;
            for (int j = 0; j < N; ++j) xn = fmaf(x[j], x[j], xn);
            if (!(sqrtf(dx_dot) > S->absTolerance && sqrtf(xn) > sqrtf(dx_dot) * S->relTolerance)) {   /* LS:1164-1173 (Q6) */
                if (age == 0) { ret.status = 1; break; }
                age = maxAge;
                continue;
            }
        } while (ret.iterations < S->maxIterations);                         /* LS:1175 */
        ret.lambda = lambda;
    }
    for (int j = 0; j < N; ++j) x_io[j] = x[j];
    *out = ret;
    free(J); free(ybuf);
    return 0;
}
