// EXPERIMENT (round 4), NOT part of the library build: a register-resident n x n kernel for 128 < n <= 256. Correct (every
// n = 129 ... 256 test of tests/ passed with it wired into launch_solve.inc) and NOT faster than k_lm_solve<double, 0>:
// 495 us per pass at n = 256 against 385-400 (profiles/r04/solve_reg_experiment_phases.txt: load 34, ?potrf 187, ?potrs 49 x 3,
// residual 35 x 3, epilogue 15). Why, so that the next attempt starts from it: (1) the factorisation is not bound by where the
// matrix lives: 16 serial 16 x 16 diagonal factorisations (2.9 us each) + an MFMA-throughput-bound trailing update at the early
// panels (816 block updates x 4 MFMAs of 64 cycles on 4 pipes = 22 us) + two barriers a panel already make ~90 us, and 400
// bytes per lane of spills (17 blocks = 136 VGPRs of the 256 a wave has at two waves per SIMD) and an 85 KB instruction
// stream (the 17 statically indexed slots) double it; (2) a column-oriented ?potrs on distributed blocks needs two workgroup
// barriers per block step (32 steps): 49 us, slower than the row-per-thread potrs_rows it was to replace (35 us).
// What it would take: blocks owned by ROW pairs (w, 15 - w: 17 blocks a wave, z_I accumulated in registers, one barrier a
// step), LDS flags instead of barriers, the transposed (coalesced) load, 16-byte residual loads. Kept for that reader.
// solve_reg.h -- the n x n part of an LM pass for 128 < n <= 256 in fp64 (cfg 4's shape), REGISTER-RESIDENT (gfx950).
//
// Same job as k_lm_solve (solve_kernel.h): gradient test, lambda_0, P = J^T J + lambda I, ?posvx('E','L') (= what
// solveBoxQP does for an unbounded problem, /root/reference/source/mir/optim/boxcqp.d:186-219), step rounding, trial point,
// predicted reduction (least_squares.d:1053-1110, 1141-1142, 1164). Problems with a finite bound keep k_lm_solve<T, 0, true>.
//
// Why another kernel. At n = 256 the lower triangle is 136 blocks of 16 x 16 doubles = 272 KB: it does not fit the CU's
// 160 KB of LDS, and round 3's kernel kept the factor in global memory -- one 256-thread workgroup waiting ~1.5-2 us per
// dependent round trip to L2: 0.47 ms a pass (factorisation 0.19, three triangular solves 0.035 each, three residuals 0.019
// each). But a CU's REGISTER FILE is 512 KB. Here one workgroup of 1024 threads (16 waves, 128 VGPRs each) holds the whole
// lower block triangle in registers -- wave w owns blocks w, w + 16, ... of the packed lower triangle (at most 9 = 72 VGPRs)
// -- for the factorisation AND the triangular solves; LDS carries only what crosses between waves: the current panel
// (16 blocks, 32 KB, double-buffered), the inverse of the current diagonal block, and the vectors.
//
//   * storage of a block B (rows of block row I, columns of block column J) in its owner's registers: lane (c = lane & 15,
//     g = lane >> 4), register q holds B[perm(c)][4 g + q] (Mma::perm: the relabelling under which a lane group owns four
//     consecutive columns). In THAT layout a block is at once the accumulator of v_mfma_f64_16x16x4 for its own update and,
//     register s = k-step s, the A or the B operand of somebody else's: B -= L_Ik L_Jk^T is four MFMAs whose operands are
//     the registers of L_Jk and L_Ik as their owners hold them -- copied through LDS lane for lane, no shuffle anywhere;
//   * right-looking Cholesky by block columns with look-ahead. Panel k: the owner of (k, k) has factored it with the
//     identity riding along (factor_diag: X = inv(L_kk)^T, solve_lds.h) and published X; the owners of (I, k) form
//     L_Ik = A_Ik X as four MFMAs (B operand: their own registers) and publish them; every owner of (I, J), J > k, applies
//     the panel -- the owner of (k + 1, k + 1) first, which then factors it while the others finish. Two barriers a panel;
//   * ?potrs column-oriented on the register-resident factor: step k applies inv(L_kk) to z_k (owner of the diagonal
//     block; both inv(L_kk) and its transpose are kept in the storage layout), then every owner of (I, k) subtracts its
//     L_Ik w_k from z_I in LDS -- one block per block row and step, the steps separated by barriers: a fixed order, no atomics;
//   * ?porfs residuals read J^T J itself (L2-hot: this workgroup has just loaded it), transposed -- thread i walks column i,
//     the lanes of a wave read 64 consecutive doubles of a row --, with the damping and the equilibration applied on the
//     fly: P and its scaled copy are never materialised (k_lm_solve<T, 0> writes both: 20 us of its 400).
#pragma once

#include "common.h"
#include "solve_types.h"

namespace mirlsq {

// phase stamps (MIR_LSQ_VARIANT_DEBUG_SOLVE), the slots of k_lm_solve's: 0 entry, 1 blocks loaded, 2 = 3 equilibrated, 4 factored,
// 5 first ?potrs, 6 = 7 refined, 8 end; 9 / 10 the shader clock at entry / end
#define MIRLSQ_STAMP_REG(ptr, k) do { if ((ptr) && threadIdx.x == 0) (ptr)[k] = wall_clock64(); } while (0)

constexpr int kRegThreads = 512, kRegWaves = 8, kRegSlots = 17, kRegNbMax = 16, kRegN = 256;

struct RegLds {
    double P[2][kRegNbMax][256];       // panel k of the factor, storage layout: [k & 1][block row][64 q + lane]
    double XB[2][16 * 17];             // X = inv(L_kk)^T of the current diagonal block: X(a, b) at a + 17 b
    double XD[kRegNbMax][256];         // X of every diagonal block in the storage layout ([64 q + lane]): the backward sweep of ?potrs
    double z[kRegN];                   // the vector of ?potrs, solved in place
    double xv[kRegN];                  // x (residuals), dx (epilogue)
    double sv[kRegN];                  // equilibration scales
    double part[2][kRegThreads / kRegN][kRegN];      // residual sums by column range: [r | w][range][row]
    double red[2 * kRegWaves];
    int info;
    int ired[3];
};

// workgroup reductions over 16 waves (every thread gets the result)
template <typename WaveOp, typename Op>
__device__ inline double reg_reduce(double v, WaveOp wop, Op op, double* red)
{
    v = wop(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < kRegWaves; ++w) r = op(r, red[w]);
    return r;
}
__device__ inline double reg_max(double v, double* red) { return reg_reduce(v, [](double a) { return wave_max(a); }, [](double a, double b) { return a > b ? a : b; }, red); }
__device__ inline double reg_min(double v, double* red) { return reg_reduce(v, [](double a) { return wave_min(a); }, [](double a, double b) { return a < b ? a : b; }, red); }
__device__ inline double reg_sum(double v, double* red) { return reg_reduce(v, [](double a) { return wave_sum(a); }, [](double a, double b) { return a + b; }, red); }

// the row sums of a mat-vec are formed by kRegRanges threads per row (each a contiguous range of kRegRows columns): fixed order
constexpr int kRegRanges = kRegThreads / kRegN, kRegRows = kRegN / kRegRanges;
__device__ inline double part_sum(const double (*part)[kRegN], int i)
{
    double t = part[0][i];
#pragma unroll
    for (int h = 1; h < kRegRanges; ++h) t += part[h][i];
    return t;
}

// sum over the four 16-lane groups of a wave (lanes i, i + 16, i + 32, i + 48): every lane gets the total
__device__ inline double groups_sum(double v)
{
    v += wave_shfl_xor(v, 16);
    v += wave_shfl_xor(v, 32);
    return v;
}

// 16 x 16 Cholesky of a diagonal block held in the storage layout, the identity riding along (see solve_lds.h: factor_diag;
// the same arithmetic). acc: in A_kk, out L_kk (lower triangle; entries above the diagonal are not meaningful).
// inv: out X = inv(L_kk)^T, the full matrix (zeros below the diagonal, reciprocal pivots on it). Returns 0 or the 1-based
// index (within the whole matrix) of the first pivot that is not positive. One wave; k = block index, n = matrix order.
__device__ __forceinline__ int reg_factor_diag(int k, int n, Mma<double>::Acc& acc, Mma<double>::Acc& inv)
{
    using M = Mma<double>;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int r = M::perm(i);
    int bad = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) inv[q] = (r == 4 * g + q) ? 1.0 : 0.0;
    static_for<4>([&](auto jj) {
        constexpr int jb = decltype(jj)::value;
        if (g == jb) {
            static_for<4>([&](auto tt) __attribute__((always_inline)) {
                constexpr int t = decltype(tt)::value;
                constexpr int c = 4 * jb + t;
                const double piv = dpp_row_bcast<M::perm(c)>(acc[t]);
                if (!(piv > 0)) { if (16 * k + c < n && bad == 0) bad = 16 * k + c + 1; }
                double rinv, d;
                rsqrt_sqrt(piv > 0 ? piv : 1.0, rinv, d);
                acc[t] = r > c ? acc[t] * rinv : (r == c ? d : acc[t]);
                inv[t] = inv[t] * rinv;
                static_for<4>([&](auto uu) {
                    constexpr int t2 = decltype(uu)::value;
                    if constexpr (t2 > t) {
                        const double l = dpp_row_bcast<M::perm(4 * jb + t2)>(acc[t]);   // L[c2][c]
                        acc[t2] -= acc[t] * l;
                        inv[t2] -= inv[t] * l;
                    }
                });
            });
        }
        if constexpr (jb < 3) {
            const double x = group_pick<jb>(acc[0], acc[1], acc[2], acc[3]);      // lane (i, t) <- L[perm(i)][4 jb + t]
            const double xb = group_pick<jb>(inv[0], inv[1], inv[2], inv[3]);     // lane (i, t) <- X[perm(i)][4 jb + t]
            const double v = r > 4 * jb + 3 ? x : 0.0;
            acc = M::mma(-v, v, acc);
            inv = M::mma(-v, xb, inv);
        }
    });
    // the groups see different pivots: the smallest bad index of the wave (0 = none)
    int b = bad ? bad : 0x7fffffff;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) { const int t = __shfl_xor(b, o, kWave); b = t < b ? t : b; }
    return b == 0x7fffffff ? 0 : b;
}

template <bool kDummy = true>
__global__ __launch_bounds__(kRegThreads) void k_lm_solve_reg(LmSolveArgs<double> a)
{
    using M = Mma<double>;
    using Acc = M::Acc;
    extern __shared__ __attribute__((aligned(16))) unsigned char reg_smem[];
    RegLds& sm = *reinterpret_cast<RegLds*>(reg_smem);
    const int n = a.n, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = lane & 15, g = lane >> 4, pr = M::perm(ci);
    const int kc = blockIdx.x;
    const int nbl = (n + 15) >> 4, NT = nbl * (nbl + 1) / 2;
    const double* __restrict__ JJ = a.JJ;
    double* dx_out = a.dx + (size_t)kc * n;
    double* trial_out = a.trial + (size_t)kc * n;
    long long* dbg = a.sc[kc].dbg;

    if (a.guard && *a.guard == 0) return;
    MIRLSQ_STAMP_REG(dbg, 0);
    if (dbg && tid == 0) dbg[9] = clock64();
    // ---- prologue: gradient test (LS:1053), lambda_0 (LS:1067-1072)
    double jy_inf = 0;
    if (a.check_grad) jy_inf = reg_max(tid < n ? fabs(a.Jy[tid]) : 0.0, sm.red);
    if (a.check_grad && tid == 0 && kc == 0) a.st->jy_inf = jy_inf;
    if (a.check_grad && !(jy_inf > a.set.gradTolerance)) {
        if (tid == 0) { ChainRec<double> r{}; r.flags = kFlagGradSmall; a.rec[kc] = r; }
        return;
    }
    const double djj = tid < n ? JJ[(size_t)tid * n + tid] : 0.0;           // undamped diagonal (lambda_0, ?poequ)
    double lambda = (kc == 0 && (a.lambda_from_state || a.lambda_from_device)) ? a.st->lambda : a.lam[kc];
    if (kc == 0 && a.lambda_from_state && !(lambda >= a.set.minLambda)) {
        const double dg = tid < n ? fabs(djj) : -1.0;
        const double mx = reg_max(dg, sm.red);
        // the FIRST diagonal entry of maximum modulus, as i?amax picks it
        int cand = (tid < n && dg == mx) ? tid : 0x7fffffff;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) { const int t = __shfl_xor(cand, o, kWave); cand = t < cand ? t : cand; }
        __syncthreads();
        if (lane == 0) sm.red[wave] = (double)cand;
        __syncthreads();
        double first = sm.red[0];
#pragma unroll
        for (int w = 1; w < kRegWaves; ++w) first = sm.red[w] < first ? sm.red[w] : first;
        const int f = (int)first;
        lambda = 0.001 * JJ[(size_t)f * n + f];
        if (!(lambda >= a.set.minLambda)) lambda = 1;
        __syncthreads();
    }

    // ---- the blocks this wave owns: packed index 16 s + wave -> (I, J), J <= I
    int bI[kRegSlots], bJ[kRegSlots];
#pragma unroll
    for (int s = 0; s < kRegSlots; ++s) {
        const int b = kRegWaves * s + wave;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= b) ++I;
        bI[s] = b < NT ? I : -1;
        bJ[s] = b < NT ? b - I * (I + 1) / 2 : -1;
    }
    // ---- load P = J^T J + lambda I (identity past n) into the storage layout
    Acc S[kRegSlots];
#pragma unroll
    for (int s = 0; s < kRegSlots; ++s) {
        if (bI[s] >= 0) {
            const int gi = 16 * bI[s] + pr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gj = 16 * bJ[s] + 4 * g + q;
                const bool in = gi < n && gj < n;
                const double t = JJ[(size_t)(in ? gi : 0) * n + (in ? gj : 0)];
                S[s][q] = in ? (gi == gj ? t + lambda : t) : (gi == gj ? 1.0 : 0.0);
            }
        } else {
            S[s] = Acc{0, 0, 0, 0};
        }
    }
    MIRLSQ_STAMP_REG(dbg, 1);
    // ---- ?poequ / ?laqsy
    const double eps = Lim<double>::eps / 2, safmin = Lim<double>::min_normal;
    const double di = tid < n ? djj + lambda : 0.0;
    const double smin = reg_min(tid < n ? di : Lim<double>::inf(), sm.red);
    const double amax = reg_max(tid < n ? di : -Lim<double>::inf(), sm.red);
    bool rcequ = false;
    double si = 1;
    if (smin > 0) {
        const double scond = sqrt(smin) / sqrt(amax);
        si = tid < n ? 1.0 / sqrt(di) : 1.0;
        const double small = safmin / Lim<double>::eps, large = 1.0 / small;
        rcequ = !(scond >= 0.1 && amax >= small && amax <= large);
    }
    if (tid < kRegN) sm.sv[tid] = rcequ ? si : 1.0;
    if (tid == 0) sm.info = 0;
    __syncthreads();
    if (rcequ) {
#pragma unroll
        for (int s = 0; s < kRegSlots; ++s) {
            if (bI[s] >= 0) {
                const double sr = sm.sv[16 * bI[s] + pr];
#pragma unroll
                for (int q = 0; q < 4; ++q) S[s][q] = (sm.sv[16 * bJ[s] + 4 * g + q] * sr) * S[s][q];
            }
        }
    }
    double bi = tid < n ? -a.Jy[tid] : 0.0;                  // right-hand side of the QP's first solve: -q (QP:191)
    if (rcequ) bi = si * bi;
    MIRLSQ_STAMP_REG(dbg, 2);
    MIRLSQ_STAMP_REG(dbg, 3);

    // ---- ?potrf: right-looking by block columns. Iteration k: the owner of (k, k) applies panel k - 1 to that block FIRST and
    //      factors it (ONE call site of the factorisation: inlined per slot it made a 390 KB kernel, six times the instruction
    //      cache), then every wave applies panel k - 1 to the rest of its blocks; barrier; the owners of (I, k) form and publish
    //      L_Ik; barrier. (The diagonal owner's other updates sit behind its factorisation: the look-ahead of solve_lds.h.)
    auto update = [&](auto ss, int par) __attribute__((always_inline)) {
        constexpr int s = decltype(ss)::value;
        const double* pj = sm.P[par][bJ[s]];
        const double* pi = sm.P[par][bI[s]];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) S[s] = M::mma(-pj[64 * s4 + lane], pi[64 * s4 + lane], S[s]);
    };
    for (int k = 0; k < nbl; ++k) {
        const int par = k & 1, pprev = (k - 1) & 1;
        // (k, k): out of its slot, panel k - 1 applied, factored, inv(L_kk) back into the slot
        bool mine = false;
        Acc acc = Acc{0, 0, 0, 0};
        static_for<kRegSlots>([&](auto ss) __attribute__((always_inline)) {
            constexpr int s = decltype(ss)::value;
            if (bI[s] == k && bJ[s] == k) { acc = S[s]; mine = true; }
        });
        if (mine) {                                          // wave-uniform
            if (k > 0) {
                const double* pk = sm.P[pprev][k];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = M::mma(-pk[64 * s4 + lane], pk[64 * s4 + lane], acc);
            }
            Acc inv;
            const int bad = reg_factor_diag(k, n, acc, inv);
            if (bad != 0 && lane == 0) atomicMax(&sm.info, 0x40000000 - bad);
            double* xb = sm.XB[par];
#pragma unroll
            for (int q = 0; q < 4; ++q) xb[pr + 17 * (4 * g + q)] = inv[q];
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = xb[(4 * g + q) + 17 * pr];     // S(inv(L_kk)): lane (c, g), register q <- X[4 g + q][perm(c)]
#pragma unroll
            for (int q = 0; q < 4; ++q) sm.XD[k][64 * q + lane] = inv[q];       // inv(L_kk)^T for the backward sweep (16 registers a wave saved)
            static_for<kRegSlots>([&](auto ss) __attribute__((always_inline)) {
                constexpr int s = decltype(ss)::value;
                if (bI[s] == k && bJ[s] == k) S[s] = acc;
            });
        }
        if (k > 0) {                                         // panel k - 1 on everything else that is still live
            static_for<kRegSlots>([&](auto ss) __attribute__((always_inline)) {
                constexpr int s = decltype(ss)::value;
                if (bJ[s] >= k && !(bI[s] == k && bJ[s] == k)) update(ss, pprev);
            });
        }
        __syncthreads();                                     // X of block k is published; panel k - 1 has been applied everywhere
        if (sm.info != 0) break;
        if (k + 1 >= nbl) break;
        // ---- rows below the diagonal block: L_Ik = A_Ik X (four MFMAs, B operand = the owner's registers), published
        {
            const double* xb = sm.XB[par];
            double xa[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) xa[s4] = xb[(4 * g + s4) + 17 * pr];     // A operand: X[4 g + s][perm(rho)]
            static_for<kRegSlots>([&](auto ss) __attribute__((always_inline)) {
                constexpr int s = decltype(ss)::value;
                if (bJ[s] == k && bI[s] > k) {
                    Acc d = Acc{0, 0, 0, 0};
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) d = M::mma(xa[s4], S[s][s4], d);
                    S[s] = d;
                    double* p = sm.P[par][bI[s]];
#pragma unroll
                    for (int q = 0; q < 4; ++q) p[64 * q + lane] = d[q];
                }
            });
        }
        __syncthreads();
    }
    __syncthreads();
    const int info = sm.info;                                // uniform
    MIRLSQ_STAMP_REG(dbg, 4);
    int qp = info != 0 ? 1 : 0;                              // QP:212-213: the unconstrained solve failed -> numericError

    // ---- ?potrs on the register-resident factor: z (LDS) <- inv(L L^T) z
    auto potrs = [&]() __attribute__((always_inline)) {
        // forward: L w = z, by block columns
        for (int k = 0; k < nbl; ++k) {
            static_for<kRegSlots>([&](auto ss) __attribute__((always_inline)) {             // w_k = inv(L_kk) z_k: per lane 4 FMAs + the sum over the lane groups
                constexpr int s = decltype(ss)::value;
                if (bI[s] == k && bJ[s] == k) {
                    double t = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) t += S[s][q] * sm.z[16 * k + 4 * g + q];
                    t = groups_sum(t);
                    wave_lds_fence();
                    if (g == 0) sm.z[16 * k + pr] = t;
                }
            });
            __syncthreads();
            if (k + 1 >= nbl) break;
            static_for<kRegSlots>([&](auto ss) __attribute__((always_inline)) {             // z_I -= L_Ik w_k
                constexpr int s = decltype(ss)::value;
                if (bJ[s] == k && bI[s] > k) {
                    double t = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) t += S[s][q] * sm.z[16 * k + 4 * g + q];
                    t = groups_sum(t);
                    if (g == 0) sm.z[16 * bI[s] + pr] -= t;
                }
            });
            __syncthreads();
        }
        // backward: L^T x = w, by block rows from the last
        for (int k = nbl - 1; k >= 0; --k) {
            static_for<kRegSlots>([&](auto ss) __attribute__((always_inline)) {             // x_k = inv(L_kk)^T w_k = X w_k
                constexpr int s = decltype(ss)::value;
                if (bI[s] == k && bJ[s] == k) {
                    double t = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) t += sm.XD[k][64 * q + lane] * sm.z[16 * k + 4 * g + q];
                    t = groups_sum(t);
                    wave_lds_fence();
                    if (g == 0) sm.z[16 * k + pr] = t;
                }
            });
            __syncthreads();
            if (k == 0) break;
            static_for<kRegSlots>([&](auto ss) __attribute__((always_inline)) {             // z_J -= L_kJ^T x_k for every block (k, J), J < k: column sums
                constexpr int s = decltype(ss)::value;
                if (bI[s] == k && bJ[s] < k && bJ[s] >= 0) {
                    const double xr = sm.z[16 * k + pr];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double t = sum16(S[s][q] * xr);   // sum over the 16 rows of the block: every lane of the group has it
                        if (ci == 0) sm.z[16 * bJ[s] + 4 * g + q] -= t;
                    }
                }
            });
            __syncthreads();
        }
    };
    // residual of the (equilibrated) system at the x in sm.xv: r = b - A x, w = |b| + |A| |x| for the row of thread tid < n.
    // A(i, j) = s_i s_j (JJ(i, j) + lambda [i == j]); thread (i = tid & 255, h = tid >> 8) walks rows j of range h (kRegN / kRegRanges rows) down
    // column i (JJ is symmetric: row j, lanes along i: coalesced)
    auto residual = [&](double b_i, double& r_i, double& w_i) __attribute__((always_inline)) {
        const int i = tid & 255, h = tid >> 8;
        double ra = 0, wa = 0;
        if (i < n) {
            const double sci = sm.sv[i];
            const int j0 = kRegRows * h, j1 = (j0 + kRegRows < n) ? j0 + kRegRows : n;
            for (int j = j0; j < j1; j += 8) {
                double av[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int jj = j + u < j1 ? j + u : j1 - 1; av[u] = JJ[(size_t)jj * n + i]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (j + u < j1) {
                        double e = (j + u == i) ? av[u] + lambda : av[u];
                        e = (sm.sv[j + u] * sci) * e;
                        const double xj = sm.xv[j + u];
                        ra += e * xj;
                        wa += fabs(e) * fabs(xj);
                    }
                }
            }
        }
        sm.part[0][h][i] = ra;
        sm.part[1][h][i] = wa;
        __syncthreads();
        if (tid < kRegN) {
            r_i = b_i - part_sum(sm.part[0], tid);
            w_i = fabs(b_i) + part_sum(sm.part[1], tid);
        }
        __syncthreads();
    };

    // ---- ?potrs, then ?porfs (iterative refinement, ITMAX = 5): ONE call site of the solve and of the residual --
    //      round 0 solves for x, round c > 0 for the correction of the residual round c - 1 left in z
    double x = 0;
    if (qp == 0) {
        const double safe1 = (double)(n + 1) * safmin, safe2 = safe1 / eps;
        double lstres = 3;
        if (tid < kRegN) sm.z[tid] = tid < n ? bi : 0.0;
        for (int count = 0;; ++count) {
            __syncthreads();
            potrs();
            if (tid < kRegN) x = count == 0 ? sm.z[tid] : x + sm.z[tid];
            if (count == 0) MIRLSQ_STAMP_REG(dbg, 5);
            if (tid < kRegN) sm.xv[tid] = tid < n ? x : 0.0;
            __syncthreads();
            double ri = 0, wi = 0;
            residual(bi, ri, wi);
            double qv = 0;
            if (tid < n) qv = (wi > safe2) ? fabs(ri) / wi : (fabs(ri) + safe1) / (wi + safe1);
            const double berr = reg_max(qv, sm.red);
            if (!(berr > eps && 2 * berr <= lstres && count + 1 <= 5)) break;
            lstres = berr;
            if (tid < kRegN) sm.z[tid] = tid < n ? ri : 0.0;
        }
        if (rcequ) x = si * x;
    }
    MIRLSQ_STAMP_REG(dbg, 6);
    MIRLSQ_STAMP_REG(dbg, 7);

    // ---- QP:216-219 (unbounded: feasible unless NaN), then LS:1087-1110, 1141-1142, 1164
    int flags = 0;
    double ndd = 0, pred = 0, xn = 0;
    if (qp == 0) {
        // a NaN in the unconstrained solution fails `l <= x <= u`: the reference enters its active-set loop, classifies every
        // variable as free and leaves with s == n (QP:265, quirk Q8) -> a status other than solved -> numericError (LS:1080)
        const int bad = reg_max((tid < n && !(x <= x)) ? 1.0 : 0.0, sm.red) > 0;
        if (bad) qp = 1;
    }
    if (qp == 0) {
        double d = 0, tr = 0;
        int moved = 0;
        if (tid < n) {
            d = x;
            const double xi = a.x[tid];
            d = d + xi;                                              // LS:1096
            d = d - xi;                                              // LS:1097
            dx_out[tid] = d;
            tr = fmax(fmin(d + xi, a.upper[tid]), a.lower[tid]);     // LS:1108-1110
            trial_out[tid] = tr;
            if (!(tr <= tr)) flags |= kFlagXNaN;
            moved = !(tr == xi);
        }
        // predicted reduction with the UNDAMPED J^T J, LS:1141-1142: t = JJ dx + 2 Jy ; pred = -(t . dx)
        __syncthreads();
        if (tid < kRegN) sm.xv[tid] = tid < n ? d : 0.0;
        __syncthreads();
        double ti = 0;
        {
            const int i = tid & 255, h = tid >> 8;
            double ra = 0;
            if (i < n) {
                const int j0 = kRegRows * h, j1 = (j0 + kRegRows < n) ? j0 + kRegRows : n;
                for (int j = j0; j < j1; j += 8) {
                    double av[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int jj = j + u < j1 ? j + u : j1 - 1; av[u] = JJ[(size_t)jj * n + i]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (j + u < j1) ra += av[u] * sm.xv[j + u];
                }
            }
            sm.part[0][h][i] = ra;
            __syncthreads();
            if (tid < n) {
                ti = part_sum(sm.part[0], tid);
                ti = ti + 2 * a.Jy[tid];
                ti = ti * d;
            }
        }
        ndd = reg_sum(d * d, sm.red);                                // LS:1099
        pred = -reg_sum(ti, sm.red);
        const double amx = reg_max(fabs(tr), sm.red);
        const int any_nan = reg_max((flags & kFlagXNaN) ? 1.0 : 0.0, sm.red) > 0;
        const int any_moved = reg_max(moved ? 1.0 : 0.0, sm.red) > 0;
        flags = any_nan ? kFlagXNaN : 0;
        if (!any_moved) flags |= kFlagNullStep;
        double sc2 = 0;                                              // ||trial||_2 for the relTolerance test, LS:1164 (scaled like ?nrm2)
        if (tid < n && amx > 0) { const double v = tr / amx; sc2 = v * v; }
        xn = amx > 0 ? amx * sqrt(reg_sum(sc2, sm.red)) : 0.0;
        if (!(sqrt(ndd) < a.set.maxStep)) flags |= kFlagStepTooLong; // LS:1101
    }
    MIRLSQ_STAMP_REG(dbg, 8);
    if (dbg && tid == 0) dbg[10] = clock64();
    if (tid == 0) {
        ChainRec<double> r{};
        r.lambda = lambda; r.new_dx_dot = ndd; r.predicted = pred; r.trial_xnorm = xn;
        r.qp_status = qp; r.qp_iterations = 0; r.flags = flags;
        a.rec[kc] = r;
    }
}

}  // namespace mirlsq
