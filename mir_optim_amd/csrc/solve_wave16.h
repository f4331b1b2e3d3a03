// solve_wave16.h -- the n x n part of an LM pass for n <= 16 on ONE wavefront, every matrix held ONE ROW PER LANE, no LDS and
// no barrier: lambda_0, P = J^T J + lambda I, solveBoxQP -- ?posvx('E','L') restated (?poequ / ?laqsy, Cholesky, ?potrs, ?porfs
// with ITMAX 5 and LAPACK's berr criterion; ?pocon and ferr only feed outputs the reference ignores, boxcqp.d:212, 323) and the
// BOXCQP active-set loop with its Kahan-Babuska-Neumaier sums (boxcqp.d:234-376) --, step rounding, trial point, predicted
// reduction (/root/reference/source/mir/optim/least_squares.d:1053-1110, 1141-1142, 1164; boxcqp.d:122-379).
//
// The f64 counterpart of posvx_rows (batched_kernel.h). Lane l = 16 g + r: row r of the matrices of GROUP g. The four 16-lane
// groups of the wave (the DPP rows) solve with FOUR damping values at once -- lambda and the three values the rejection rule
// (LS:1103, 1127: lambda *= lambdaIncrease mu, mu *= 2) makes of it next -- at the cost of one solve: a rejected pass finds its
// step ready. Values move between the lanes of a group by DPP row_newbcast / row_ror (VALU lane moves). A ladder level above
// the first is only offered when its unconstrained solution is feasible; when the FIRST level's is not, the whole wave runs
// that one system's active-set loop (all groups on the same data: uniform control flow).
//
// Reduced systems of the active-set loop are solved in place at full size: a bound variable's row and column are replaced by
// the identity's, which leaves the arithmetic of the free part exactly that of the compact s x s system in the reference's
// order (every skipped term is an exact zero), and ?poequ looks at the free rows only.
#pragma once

#include "common.h"
#include "solve_types.h"

namespace mirlsq {

constexpr int kW16 = 16;

__device__ __forceinline__ double row16_max(double v)
{
    v = fmax(v, dpp_row_ror<8>(v)); v = fmax(v, dpp_row_ror<4>(v)); v = fmax(v, dpp_row_ror<2>(v)); v = fmax(v, dpp_row_ror<1>(v));
    return v;
}
__device__ __forceinline__ double row16_min(double v)
{
    v = fmin(v, dpp_row_ror<8>(v)); v = fmin(v, dpp_row_ror<4>(v)); v = fmin(v, dpp_row_ror<2>(v)); v = fmin(v, dpp_row_ror<1>(v));
    return v;
}
__device__ __forceinline__ double row16_sum(double v) { return sum16(v); }
// this group's 16 bits of a wave ballot
__device__ __forceinline__ unsigned group_bits(bool pred, int g) { return (unsigned)((__ballot(pred) >> (16 * g)) & 0xffffull); }

// ?posvx('E','L') of the group's system (M + shift I) x = rhs: Mrow = row r of the symmetric M; rows with live == false (and
// every row >= N) are identity rows. Returns info (group-uniform: 0, or the 1-based index of the first non-positive pivot); x_r out.
template <int N>
__device__ __forceinline__ int posvx_rows16(const double (&Mrow)[kW16], double shift, double d_r, double rhs_r, bool live, int r, double& x_r,
                                            int order = N)
{
    // The matrix is M + shift I, row r in Mrow (no damped copy is kept: registers). d_r = Mrow[r] + shift, handed in by the
    // caller: picking it out of the register array with a run-time index would put the array in scratch memory
    const double eps = Lim<double>::eps / 2, safmin = Lim<double>::min_normal;
    // ?poequ over the live rows
    const double smin = row16_min(live ? d_r : Lim<double>::inf());
    const double amax = row16_max(live ? d_r : -Lim<double>::inf());
    const bool pos = smin > 0;
    const double scond = sqrt(smin) / sqrt(amax);
    const double s_r = (pos && live) ? 1.0 / sqrt(d_r) : 1.0;
    const double small = safmin / Lim<double>::eps, large = 1.0 / small;
    const bool rcequ = pos && !(scond >= 0.1 && amax >= small && amax <= large);
    // ?laqsy; identity rows / columns for what is not live
    double Arow[kW16], Frow[kW16], Fcol[kW16];
    static_for<kW16>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const double sk = dpp_row_bcast<k>(s_r);
        const bool lk = dpp_row_bcast<k>(live ? 1 : 0) != 0;
        const double v = (r == k) ? Mrow[k] + shift : Mrow[k];
        Arow[k] = (live && lk && k < N) ? (rcequ ? sk * s_r * v : v) : (r == k ? 1.0 : 0.0);
        Frow[k] = Arow[k];
        Fcol[k] = 0.0;
    });
    const double b_r = live ? (rcequ ? s_r * rhs_r : rhs_r) : 0.0;
    // ?potrf 'L', right-looking by columns: after step j, Frow[jj] (jj > j) of row r >= jj holds A[r][jj] - sum_{k <= j} L[r][k] L[jj][k]
    int info = 0;
    double rdiag = 1.0;                                          // 1 / L[r][r]
    static_for<kW16>([&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < N) {
            const double ajj = dpp_row_bcast<j>(Frow[j]);
            info = (info == 0 && !(ajj > 0)) ? j + 1 : info;
            double rinv, d;
            rsqrt_sqrt(ajj > 0 ? ajj : 1.0, rinv, d);
            Frow[j] = (r == j) ? d : Frow[j] * rinv;             // rows above the diagonal carry values nobody reads
            rdiag = (r == j) ? rinv : rdiag;
            static_for<kW16>([&](auto JJ) {
                constexpr int jj = decltype(JJ)::value;
                if constexpr (jj > j && jj < N) {
                    const double ljj = dpp_row_bcast<jj>(Frow[j]);        // L[jj][j]
                    Frow[jj] = fma(-Frow[j], ljj, Frow[jj]);
                    Fcol[jj] = (r == j) ? ljj : Fcol[jj];                 // lane j collects column j of L
                }
            });
        }
    });
    // ?potrs with the vector distributed (component r in lane r): L y = v by columns, then L^T z = y by columns of L^T
    auto potrs = [&](double v) {
        static_for<kW16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i < N) {
                const double yi = dpp_row_bcast<i>(v * rdiag);
                v = (r == i) ? yi : (r > i ? fma(-Frow[i], yi, v) : v);
            }
        });
        static_for<kW16>([&](auto II) {
            constexpr int i = kW16 - 1 - decltype(II)::value;
            if constexpr (i < N) {
                const double zi = dpp_row_bcast<i>(v * rdiag);
                v = (r == i) ? zi : (r < i ? fma(-Fcol[i], zi, v) : v);
            }
        });
        return v;
    };
    double x = potrs(b_r);
    // ?porfs: the loop runs while any group refines; a group that has stopped keeps its solution
    const double safe1 = (double)(order + 1) * safmin, safe2 = safe1 / eps;     // order: rows of the system (<= N)
    double lstres = 3;
    bool active = true;
    for (int count = 1;; ++count) {
        double ri = b_r, wi = fabs(b_r);
        static_for<kW16>([&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (k < N) {
                const double xk = dpp_row_bcast<k>(x);
                ri = fma(-Arow[k], xk, ri);
                wi = fma(fabs(Arow[k]), fabs(xk), wi);
            }
        });
        const bool big = wi > safe2;
        const double q = (big ? fabs(ri) : fabs(ri) + safe1) / (big ? wi : wi + safe1);
        const double berr = row16_max((live && r < N) ? q : 0.0);
        active = active && berr > eps && 2 * berr <= lstres && count <= 5;
        if (__ballot(active) == 0) break;
        const double c = potrs((live && r < N) ? ri : 0.0);
        x = active ? x + c : x;
        lstres = active ? berr : lstres;
    }
    x_r = rcequ ? s_r * x : x;
    return info;
}

// what one ladder level hands to the acceptance logic (group-uniform values; dx_r / trial_r per lane)
struct Wave16Level {
    double lambda, ndd, pred, xnorm;
    int qp_status, qp_iters, flags;
    bool offered;                        // the level may be used by a later pass (always true for level 0)
};

// One pass's n x n work for the ladder lam[0..3] (lam_g: this lane's group value; group 0 = the pass itself).
// JJrow: row r of the UNDAMPED J^T J (full symmetric), djj = JJrow[r]; Jy_r, x_r, lower_r, upper_r: component r (r >= N: anything).
// check_grad: LS:1053 first (returns with kFlagGradSmall in every level's flags when |J^T y|_inf <= gradTolerance);
// from_state: apply the lambda_0 rule LS:1067-1072 (the ladder is then rebuilt from lambda_0 with inc, mu).
template <int N, bool BOUNDED>
__device__ __forceinline__ void wave16_lm_solve(const double (&JJrow)[kW16], double djj, double Jy_r, double x_r, double lo_r, double up_r,
                                                double lambda, double mu, bool check_grad, bool from_state,
                                                const LmSettingsDev<double>& set, Wave16Level& out, double& dx_out, double& trial_out,
                                                int n = N, bool single = false)
{
    // n <= N: the problem's order at run time (rows n .. N - 1 are identity rows: the launch-chain kernel k_lm_solve_wave is
    // compiled once, for N = 16); single: no ladder, every group solves with `lambda` (one ladder entry per wave)
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const bool el = r < n;
    out.lambda = lambda; out.ndd = 0; out.pred = 0; out.xnorm = 0; out.qp_status = 0; out.qp_iters = 0; out.flags = 0; out.offered = g == 0;
    dx_out = 0; trial_out = x_r;
    if (check_grad) {                                            // LS:1053
        const double jy_inf = row16_max(el ? fabs(Jy_r) : 0.0);
        if (!(jy_inf > set.gradTolerance)) { out.flags = kFlagGradSmall; return; }
    }
    if (from_state && !(lambda >= set.minLambda)) {              // LS:1067-1072: the FIRST entry of maximum |diag|, as i?amax picks it
        const double ad = el ? fabs(djj) : -1.0;
        const double mx = row16_max(ad);
        const unsigned hit = group_bits(el && ad == mx, g);
        const int first = hit ? __builtin_ctz(hit) : 0;
        const double dfirst = __shfl(djj, 16 * g + first, 64);
        lambda = 0.001 * dfirst;
        if (!(lambda >= set.minLambda)) lambda = 1;
    }
    // the ladder: group g solves with the damping g rejections from now would leave (LS:1103-1104, 1127-1128)
    double lam_g = lambda;
    {
        double l = lambda, mm = mu;
#pragma unroll
        for (int k = 1; k < 4; ++k) { l *= set.lambdaIncrease * mm; mm *= 2; lam_g = (g == k && !single) ? l : lam_g; }
    }
    out.lambda = lam_g;
    const double qpl = lo_r - x_r, qpu = up_r - x_r;              // LS:1074-1077
    // P = J^T J + lambda I, LS:1079 (quirk Q1), is never formed: the solves take J^T J and the shift
    // ---- solveBoxQP, boxcqp.d:122-379
    double xq;
    int qp = 0, qp_iters = 0;
    {
        const int info = posvx_rows16<N>(JJrow, lam_g, djj + lam_g, -Jy_r, el, r, xq, n);                          // QP:168-214
        if (info != 0) qp = 1;
    }
    bool infeasible = el && !(qpl <= xq && xq <= qpu);                                       // QP:216-219 (NaN counts)
    const unsigned inf_bits = group_bits(infeasible, g);
    if (g > 0 && (inf_bits != 0 || qp != 0)) out.offered = false;
    else if (g > 0) out.offered = true;
    if constexpr (BOUNDED) {
        const unsigned inf0 = (unsigned)(__ballot(infeasible) & 0xffffull);
        const int qp0 = __builtin_amdgcn_readlane(qp, 0);
        if (qp0 == 0 && inf0 != 0) {
            // the pass itself needs the active-set loop: every group takes level 0's system (uniform control flow from here on;
            // the other levels are not offered)
            const double lam0 = lane_bcast(lam_g, 0);
            auto P0 = [&](auto K) { constexpr int k = decltype(K)::value; return (r == k) ? JJrow[k] + lam0 : JJrow[k]; };   // row r of P
            double x = __shfl(xq, r, 64);
            const double q_r = Jy_r;
            double la = 0, mul = 0;                                                          // QP:228-232
            uint32_t maxit = set.qpMaxIterations ? set.qpMaxIterations : (uint32_t)n * 10 + 100;   // QP:224-226
            int st0 = 2;                                                                      // QP:378
            for (uint32_t step = 0; step < maxit; ++step) {                                   // QP:234
                qp_iters = (int)step + 1;
                int fl = 2;                                                                   // 2 = not an element
                if (el) {                                                                     // QP:239-263
                    const double xl = x - qpl, ux = qpu - x;
                    if (xl < 0 || (xl < set.qpRelTolerance + set.qpAbsTolerance * fabs(qpl) && la >= 0)) { fl = -1; x = qpl; mul = 0; }
                    else if (ux < 0 || (ux < set.qpRelTolerance + set.qpAbsTolerance * fabs(qpu) && mul >= 0)) { fl = 1; x = qpu; la = 0; }
                    else { fl = 0; mul = 0; la = 0; }
                }
                const unsigned free_bits = (unsigned)(__ballot(fl == 0) & 0xffffull);
                const int sN = __builtin_popcount(free_bits);
                if (sN == n) break;                                                           // QP:265-266 (quirk Q8)
                // right-hand side of the reduced system, QP:282-305: Kahan-Babuska-Neumaier over the bound variables, j ascending
                double ks = q_r, kc = 0;
                static_for<kW16>([&](auto JX) {
                    constexpr int j = decltype(JX)::value;
                    if constexpr (j < N) {
                        const int fj = dpp_row_bcast<j>(fl);
                        const double bj = dpp_row_bcast<j>(x);                               // a bound variable sits ON its bound
                        const double v = P0(JX) * bj;
                        const double t = ks + v;
                        const double kn = (fabs(ks) >= fabs(v)) ? kc + ((ks - t) + v) : kc + ((v - t) + ks);
                        kc = fj ? kn : kc;
                        ks = fj ? t : ks;
                    }
                });
                const double b_r = -(ks + kc);
                if (sN) {                                                                     // QP:307-329
                    double xs;
                    const int info = posvx_rows16<N>(JJrow, lam0, djj + lam0, b_r, fl == 0, r, xs, n);
                    if (info != 0) { st0 = 1; break; }
                    x = fl == 0 ? xs : x;
                }
                // multipliers of the bound variables, QP:333-337 (two partial sums, as the reference's two dot products)
                double v1 = 0, v2 = 0;
                static_for<kW16>([&](auto JX) {
                    constexpr int j = decltype(JX)::value;
                    if constexpr (j < N) {
                        const double xj = dpp_row_bcast<j>(x);
                        v1 = (j < r) ? fma(P0(JX), xj, v1) : v1;
                        v2 = (j >= r) ? fma(P0(JX), xj, v2) : v2;
                    }
                });
                const double val = v1 + v2 + q_r;
                la = fl == -1 ? val : la;
                mul = fl == 1 ? -val : mul;
                bool again = false;                                                           // QP:339-347
                if (fl == -1) again = !(la >= 0);
                else if (fl == 1) again = !(mul >= 0);
                else if (fl == 0) again = !(x >= qpl && x <= qpu);
                if ((__ballot(again) & 0xffffull) != 0) continue;
                x = el ? fmax(fmin(x, qpu), qpl) : x;                                         // QP:349 applyBounds
                st0 = 0;
                break;
            }
            // level 0's answer lives in every group now; only group 0 reports it
            if (g == 0) { xq = x; qp = st0; }
            infeasible = false;
        }
    } else {
        // all bounds infinite: the loop could only be entered with a NaN in the solution (quirk Q8): not `solved`
        if (inf_bits != 0 && qp == 0) qp = 1;
    }
    out.qp_status = qp; out.qp_iters = qp_iters;
    // ---- LS:1087-1110, 1141-1142, 1164
    int flags = 0;
    double d = el ? xq : 0.0;
    const bool nan_d = el && !(d <= d);                                                       // LS:1087
    d = d + x_r;                                                                              // LS:1096
    d = d - x_r;                                                                              // LS:1097
    d = el ? d : 0.0;
    const double tr = el ? fmax(fmin(d + x_r, up_r), lo_r) : 0.0;                             // LS:1108-1110
    const bool nan_t = el && !(tr <= tr);
    const bool moved = el && !(tr == x_r);                                                    // NaN counts as moved
    if (group_bits(nan_d, g)) flags |= kFlagDxNaN;
    if (group_bits(nan_t, g)) flags |= kFlagXNaN;
    if (!group_bits(moved, g)) flags |= kFlagNullStep;
    const double ndd = row16_sum(d * d);                                                      // LS:1099
    double ti = 0;                                                                            // LS:1141-1142 with the UNDAMPED J^T J
    static_for<kW16>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < N) ti = fma(JJrow[k], dpp_row_bcast<k>(d), ti);
    });
    ti = (ti + 2 * Jy_r) * d;
    const double pred = -row16_sum(el ? ti : 0.0);
    const double amx = row16_max(fabs(tr));                                                   // ||trial||_2 scaled like ?nrm2, LS:1164
    double sc2 = 0;
    if (el && amx > 0) { const double v = tr / amx; sc2 = v * v; }
    const double xn = amx > 0 ? amx * sqrt(row16_sum(sc2)) : 0.0;
    if (!(sqrt(ndd) < set.maxStep)) flags |= kFlagStepTooLong;                                // LS:1101
    out.ndd = ndd; out.pred = pred; out.xnorm = xn; out.flags = flags;
    dx_out = d; trial_out = tr;
}

// The same as an out-of-line routine on LDS-resident operands (the resident-J solver's workgroup 0 calls it from its round loop:
// out of line, the ~170 registers of the solve do not weigh on that loop's allocation). Called by ONE wave. JJ: 16 x 16 (leading
// dimension 16), Jy, xs, lo, up: 16 each; out: dx / trial of level g at dxo + 16 g / tro + 16 g, level records at lrec + 8 g
// (lambda, ndd, pred, xnorm) and lreci + 4 g (qp_status, qp_iterations, flags, offered).
template <int N, bool BOUNDED>
__device__ __forceinline__ void wave16_lm_solve_lds(const double* JJ, const double* Jy, const double* xs, const double* lo, const double* up,
                                                 double lambda, double mu, int check_grad, int from_state,
                                                 const LmSettingsDev<double>* set, double* dxo, double* tro, double* lrec, int* lreci)
{
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    double JJrow[kW16];
#pragma unroll
    for (int k = 0; k < kW16; ++k) JJrow[k] = JJ[r * kW16 + k];
    const LmSettingsDev<double> S = *set;
    Wave16Level lv;
    double d, t;
    wave16_lm_solve<N, BOUNDED>(JJrow, JJ[r * kW16 + r], Jy[r], xs[r], lo[r], up[r], lambda, mu, check_grad != 0, from_state != 0, S, lv, d, t);
    dxo[16 * g + r] = d;
    tro[16 * g + r] = t;
    if (r == 0) {
        lrec[8 * g] = lv.lambda; lrec[8 * g + 1] = lv.ndd; lrec[8 * g + 2] = lv.pred; lrec[8 * g + 3] = lv.xnorm;
        lreci[4 * g] = lv.qp_status; lreci[4 * g + 1] = lv.qp_iters; lreci[4 * g + 2] = lv.flags; lreci[4 * g + 3] = lv.offered ? 1 : 0;
    }
}

}  // namespace mirlsq
