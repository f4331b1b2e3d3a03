// solve_kernel.h -- the n x n part of one LM pass as ONE single-workgroup kernel:
//   damping (lambda_0, P = J^T J + lambda I), step bounds, the bound-constrained QP
//   (equilibrated Cholesky solve with iterative refinement == LAPACK ?posvx('E','L'), then the
//   BOXCQP active-set loop when a bound is violated), step rounding, trial point, predicted
//   reduction and the guards of the reference.
//
// Replaces (/root/reference/source/mir/optim/):
//   least_squares.d:1067-1079, 1087-1110, 1141-1142, 1164 (nrm2)      -> k_lm_solve
//   boxcqp.d:122-379 (solveBoxQP), 404-410 (applyBounds)              -> box_qp_device
//   mir-lapack posvx (un-vendored; Netlib ?posvx = ?poequ + ?laqsy + ?potrf + ?potrs + ?porfs)
//                                                                     -> posvx_device
// The reciprocal-condition estimate (?pocon) and the forward-error bound of ?porfs are not
// computed: they only feed rcond/ferr and `info == n + 1`, which the reference ignores
// (boxcqp.d:212, 323), so no result depends on them.
//
// MI355X mapping: one 256-thread workgroup; the Cholesky factor lives in LDS, column-major with an
// odd leading dimension (both column and row walks are bank-conflict-free) when it fits
// (n <= 128 in f64), else in an L2-resident global scratch; triangular solves run on one wave with
// the vector in registers (no workgroup barrier on the critical path).
#pragma once

#include "common.h"
#include "misc_kernels.h"
#include "solve_types.h"

namespace mirlsq {

#define MIRLSQ_STAMP(ptr, k) do { if ((ptr) && threadIdx.x == 0) (ptr)[k] = wall_clock64(); } while (0)

// ---------------------------------------------------------------- workgroup collectives
// wave stage on DPP row operations (+ two cross-row shuffles), then one LDS exchange between the four waves
template <typename T, typename WaveOp, typename Op>
__device__ inline T block_reduce(T v, WaveOp wop, Op op, T* red /* >= 4 */)
{
    v = wop(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    T r = red[0];
#pragma unroll
    for (int w = 1; w < kSolveThreads / kWave; ++w) r = op(r, red[w]);
    return r;
}
template <typename T> __device__ inline T block_sum(T v, T* red)
{
    return block_reduce(v, [](T a) { return wave_sum(a); }, [](T a, T b) { return a + b; }, red);
}
template <typename T> __device__ inline T block_max(T v, T* red)
{
    return block_reduce(v, [](T a) { return wave_max(a); }, [](T a, T b) { return a > b ? a : b; }, red);
}
template <typename T> __device__ inline T block_min(T v, T* red)
{
    return block_reduce(v, [](T a) { return wave_min(a); }, [](T a, T b) { return a < b ? a : b; }, red);
}
__device__ inline int block_or(int v, int* red)
{
    __syncthreads();
    if (threadIdx.x == 0) *red = 0;
    __syncthreads();
    if (v) atomicOr(red, v);
    __syncthreads();
    return *red;
}

}  // namespace mirlsq
#include "solve_lds.h"
namespace mirlsq {

// The out-of-line routines below receive generic pointers; dereferenced as such they become FLAT loads, which count on both
// memory counters -- every "wait for my LDS operations" before a barrier then also waits for global loads that were issued
// ahead on purpose. as_global / as_lds name the address space again (the callers only pass global scratch / __shared__ arrays).
template <typename T> using gbl_cptr = const __attribute__((address_space(1))) T*;
template <typename T> using gbl_ptr = __attribute__((address_space(1))) T*;
template <typename T> using lds_ptr = __attribute__((address_space(3))) T*;
template <typename T> __device__ __forceinline__ gbl_cptr<T> as_global(const T* p) { return (gbl_cptr<T>)p; }
template <typename T> __device__ __forceinline__ gbl_ptr<T> as_global_w(T* p) { return (gbl_ptr<T>)p; }
template <typename T> __device__ __forceinline__ lds_ptr<T> as_lds(T* p) { return (lds_ptr<T>)p; }

// ---------------------------------------------------------------- generic path helpers (factor in global memory)
// Inverses of the 16 x 16 diagonal blocks of L (lower triangular), one thread per (block, column):
// column c of inv(L_kk) is the forward substitution L_kk x = e_c. Dinv block k at Dinv + k * 272,
// element (r, c) at [r + 17 c] (leading dimension 17: row- and column-wise walks are conflict-free).
template <typename T, int NB>
__device__ __forceinline__ void invert_diag_blocks(int n, const T* F, int ldf, T* Dinv)
{
    const int tid = threadIdx.x;
    if (tid < 16 * NB) {
        const int k = tid >> 4, c = tid & 15;
        const int base = 16 * k;
        T x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            T s = (r == c) ? T(1) : T(0);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < r) {
                    const int ir = base + r < n ? base + r : n - 1, iq = base + q < n ? base + q : n - 1;
                    s -= F[ir + (size_t)iq * ldf] * x[q];
                }
            const int ir = base + r < n ? base + r : n - 1;
            const T dr = F[ir + (size_t)ir * ldf];
            // rows past n (partial last block) are treated as an identity extension
            x[r] = (base + r < n) ? ((r >= c) ? s / dr : T(0)) : ((r == c) ? T(1) : T(0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Dinv[k * 272 + r + 17 * c] = x[r];
    }
    __syncthreads();
}

// ---------------------------------------------------------------- generic path (factor in global memory), 128 < n <= 256
// potrf_panel: left-looking Cholesky by 16-column panels (n <= kSolveThreads = 256), three barriers per panel:
//   1. S = L[:, :16k] L[16k:16k+16, :16k]^T, the contribution of every earlier panel to panel k, on MFMA 16x16x4:
//      wave w owns the four 16-row blocks of rows 64 w .. 64 w + 63; per earlier panel j and k-step it loads one
//      B fragment (the 16 panel rows) and one A fragment per row block straight from the factor (global,
//      L2-resident; the loads of panel j + 1 are issued before the MFMAs of panel j) and S goes through LDS;
//   2. ONE ROW PER THREAD from here on: thread i takes p[c] = A[i][16k + c] - S[i][c] into 16 registers; the wave
//      that holds the 16 diagonal rows factors them wave-synchronously (16 pivots, v_readlane broadcasts, no barrier)
//      and publishes L_kk and the reciprocal pivots through LDS; the same loop finishes the other rows of that wave;
//   3. every other row solves its 16 entries against L_kk (row-wise forward substitution); the panel is stored.
// History at n = 256: column-by-column through global memory 3.4 ms per solve kernel; row-per-thread with the update as
// FMAs + v_readlane broadcasts 0.37 ms for the factorisation alone; this version moves that update to the matrix cores.
// Same pivot arithmetic as potrf_tiled2 (rsqrt + multiply). The generic-path routines are NOT inlined: each is large
// unrolled code that box_qp_device would otherwise instantiate four times per kernel; as out-of-line functions their
// pointer arguments are generic (flat loads), which this path can afford.
template <typename T>
__device__ __noinline__ int potrf_panel(int n, const T* A_, int lda, T* F_, int ldf, T* blk_, T* rd_, int* info_s, T* spanel_,
                                        long long* dbg = nullptr)
{
    using Acc = typename Mma<T>::Acc;
    const gbl_cptr<T> A = as_global(A_);
    const gbl_ptr<T> F = as_global_w(F_);
    const lds_ptr<T> blk = as_lds(blk_), rd = as_lds(rd_), spanel = as_lds(spanel_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tid;
    const int nblk = (n + 15) / 16;
    const int lr = lane & 15, lk = lane >> 4;
    if (tid == 0) *info_s = 0;
    __syncthreads();
    long long ph[4] = {0, 0, 0, 0};       // DEBUG_SOLVE: time in steps 1, 2, 3 summed over the panels (thread 0)
    for (int k = 0; k < nblk; ++k) {
        const int c0 = 16 * k;
        if (dbg && tid == 0) ph[3] = wall_clock64();
        // the panel's entries of A (row i, 16 columns) do not depend on step 1: their loads fly while the matrix cores work
        const int ic = i < n ? i : n - 1;
        T pa[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) pa[c] = A[ic + (size_t)(c0 + c < n ? c0 + c : n - 1) * lda];
        // ---- 1. S for this wave's row blocks rb = wave + 4 u that reach into the panel (rb >= k). Dealt CYCLICALLY: the
        //      blocks below the panel are the ones that have work, and with contiguous blocks per wave the last wave carried
        //      16 k MFMAs per panel while the first had none (51 us of a 220 us factorisation on one SIMD)
        if (k > 0) {
            Acc acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = Acc{0, 0, 0, 0};
            int rowa[4];
            bool live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rb = wave + 4 * u;
                const int r = 16 * rb + lr;
                rowa[u] = r < n ? r : n - 1;
                live[u] = rb >= k && 16 * rb < n;               // wave-uniform
            }
            const int rowb = c0 + lr < n ? c0 + lr : n - 1;
            // Ring of three register buffers: the fragments of TWO earlier panels in flight beyond the one on the matrix cores.
            // The loads of a panel are issued unconditionally -- a count the compiler can keep in vmcnt; with `if (live[u])` around
            // them it cannot know how many are in flight and waits for all of them before each batch of MFMAs -- so the loop is
            // compiled once per first live block UMIN (this wave's blocks UMIN .. 3 reach into the panel; blocks past n read
            // their clamped rows for nothing): no load is issued for a block above the panel.
            constexpr int kAheadF = 2, kRingF = kAheadF + 1;
            auto sweep = [&](auto UMIN_) {
                constexpr int UMIN = decltype(UMIN_)::value;
                T fa[kRingF][4][4], fb[kRingF][4];              // [buffer][k-step][row block]
                auto load = [&](int j, auto B) {
                    constexpr int buf = decltype(B)::value;
                    const int jc = j < k ? j : k - 1;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const size_t col = (size_t)(16 * jc + 4 * s4 + lk) * ldf;
                        fb[buf][s4] = F[rowb + col];
#pragma unroll
                        for (int u = UMIN; u < 4; ++u) fa[buf][s4][u] = F[rowa[u] + col];
                    }
                };
                auto mmas = [&](auto B) {
                    constexpr int buf = decltype(B)::value;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int u = UMIN; u < 4; ++u) if (live[u]) acc[u] = Mma<T>::mma(fa[buf][s4][u], fb[buf][s4], acc[u]);
                };
                static_for<kAheadF>([&](auto U) { load(decltype(U)::value, U); });
                for (int j0 = 0; j0 < k; j0 += kRingF) {
                    static_for<kRingF>([&](auto U) {
                        constexpr int u0 = decltype(U)::value;
                        if (j0 + u0 < k) {
                            load(j0 + u0 + kAheadF, IntC<(u0 + kAheadF) % kRingF>{});
                            mmas(U);
                        }
                    });
                }
            };
            const int umin = k > wave ? (k - wave + 3) / 4 : 0;     // first u with wave + 4 u >= k (wave-uniform)
            if (live[0] || live[1] || live[2] || live[3]) {
                if (umin == 0) sweep(IntC<0>{});
                else if (umin == 1) sweep(IntC<1>{});
                else if (umin == 2) sweep(IntC<2>{});
                else sweep(IntC<3>{});
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (live[u]) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            spanel[(16 * (wave + 4 * u) + Mma<T>::row(lane, r)) * 17 + lr] = acc[u][r];   // D: row, col = lane & 15
                    }
            }
        }
        __syncthreads();
        if (dbg && tid == 0) { const long long t = wall_clock64(); ph[0] += t - ph[3]; ph[3] = t; }
        // ---- 2. row per thread
        const bool active = i < n && i >= c0;
        T p[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const T v = pa[c];
            const T sv = k > 0 ? spanel[i * 17 + c] : T(0);
            p[c] = (active && c0 + c < n) ? v - sv : ((i == c0 + c) ? T(1) : T(0));   // identity padding past n
        }
        // the diagonal rows c0 .. c0 + 15 are lanes l0 .. l0 + 15 of wave wd: ONE DPP row (c0 is a multiple of 16). They factor
        // the block among themselves with row_newbcast moves (VALU; v_readlane with a run-time lane cost an SGPR round trip per
        // pivot and per multiplier: 4.9 us per panel); the other three rows of that wave execute the same instructions on
        // values nobody uses and take step 3 like every other row.
        const int wd = c0 >> 6, l0 = c0 & 63;
        const bool own = wave == wd && lane >= l0 && lane < l0 + 16;
        if (wave == wd) {
            const int il = lane & 15;
            int bad = 0;
            static_for<16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const T piv = dpp_row_bcast<c>(p[c]);
                if (!(piv > 0)) { if (c0 + c < n && bad == 0) bad = c0 + c + 1; }
                T rinv, d;
                rsqrt_sqrt(piv > 0 ? piv : T(1), rinv, d);
                if (own) { if (il > c) p[c] *= rinv; else if (il == c) p[c] = d; }
                if (own && il == c) rd[c] = rinv;
                static_for<16>([&](auto c2c) {
                    constexpr int c2 = decltype(c2c)::value;
                    if constexpr (c2 > c) {
                        const T lc = dpp_row_bcast<c2>(p[c]);    // L[c0 + c2][c0 + c]
                        if (own && il > c) p[c2] -= p[c] * lc;
                    }
                });
            });
            if (own) {
#pragma unroll
                for (int c = 0; c < 16; ++c) blk[il * 17 + c] = p[c];
            }
            if (bad != 0 && lane == l0) *info_s = bad;
        }
        __syncthreads();
        if (dbg && tid == 0) { const long long t = wall_clock64(); ph[1] += t - ph[3]; ph[3] = t; }
        const int info = *info_s;
        if (info != 0) return info;                             // uniform
        // ---- 3. the other rows against L_kk, store
        if (!own && active) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                T v = p[c];
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t < c) v -= p[t] * blk[c * 17 + t];
                p[c] = v * rd[c];
            }
        }
        if (active) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c0 + c < n && i >= c0 + c) F[i + (size_t)(c0 + c) * ldf] = p[c];
        }
        __syncthreads();                                        // the panel is visible to the later ones; blk / rd / spanel reusable
        if (dbg && tid == 0) { const long long t = wall_clock64(); ph[2] += t - ph[3]; }
    }
    if (dbg && tid == 0) { dbg[16] = ph[0]; dbg[17] = ph[1]; dbg[18] = ph[2]; }
    return 0;
}

template <typename T>
__device__ __noinline__ void invert_diag_blocks_generic(int n, const T* F, int ldf, T* Dinv) { invert_diag_blocks<T, 16>(n, F, ldf, Dinv); }
// ?potrs for the generic path, ONE ROW PER THREAD (n <= kSolveThreads): block step kb = the 16 owners of rows
// 16 kb .. 16 kb + 15 (16 lanes of one wave) form x_kb = inv(L_kk) z_kb with v_readlane broadcasts and publish it;
// after ONE barrier every remaining row subtracts its 16 products. The factor entries a row needs do not depend on
// the solution, so their loads (global, L2) are issued before the barrier. Replaces the one-wave blocked solve here
// (105 us per call at n = 256: 32 dependent block steps, each a round trip to L2).
template <typename T>
__device__ __noinline__ void potrs_rows(int n, const T* F_, int ldf, const T* Dinv_, T* xv_, T* xk_)
{
    const gbl_cptr<T> F = as_global(F_);
    const lds_ptr<T> Dinv = as_lds(const_cast<T*>(Dinv_)), xk = as_lds(xk_);
    const gbl_ptr<T> xv = as_global_w(xv_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tid, ic = i < n ? i : n - 1;
    const int nb = (n + 15) / 16;
    __syncthreads();
    T z = i < n ? xv[i] : T(0);
    // The factor entries a row needs in block step kb do not depend on the solution: the loads of step kb + 1 are issued
    // before step kb's barrier, so a step costs the diagonal solve + one barrier instead of an L2 round trip (45 -> ~15 us
    // per call at n = 256).
    // A wave none of whose rows takes part in a block step skips that step's loads (half of them on average): the ablation of
    // profiles/r04/potrs_rows_ablation_n256.txt puts 15 of a call's 40 us on ISSUING the loads -- the CU's vector-memory address rate.
    T lnext[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) lnext[c] = 0;
    auto load_row = [&](int kb) {                              // L[i][16 kb .. 16 kb + 15], used by rows i >= 16 kb + 16
        if (64 * wave + 63 >= 16 * kb + 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c) lnext[c] = F[ic + (size_t)(16 * kb + c < n ? 16 * kb + c : n - 1) * ldf];
        }
    };
    auto load_col = [&](int kb) {                              // L[16 kb .. 16 kb + 15][i], used by rows i < 16 kb
        if (64 * wave < 16 * kb) {
#pragma unroll
            for (int c = 0; c < 16; ++c) lnext[c] = F[(16 * kb + c < n ? 16 * kb + c : n - 1) + (size_t)ic * ldf];
        }
    };
    // ---- forward: L z = b
    load_row(0);
    for (int kb = 0; kb < nb; ++kb) {
        const int c0 = 16 * kb;
        T lrow[16];                                             // L[i][c0 .. c0 + 15], needed when i >= c0 + 16
#pragma unroll
        for (int c = 0; c < 16; ++c) lrow[c] = lnext[c];
        if (kb + 1 < nb) load_row(kb + 1); else load_col(nb - 1);
        if (wave == (c0 >> 6)) {
            // the 16 owners are one DPP row of the wave (c0 is a multiple of 16): the broadcasts of z are row_newbcast moves on
            // the VALU (no SGPR round trip per term as with v_readlane), the 16 coefficients are loaded in one batch
            const int l0 = c0 & 63, r = lane & 15;
            T dv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) dv[c] = Dinv[kb * 272 + r + 17 * c];                         // entries above the diagonal are 0
            T xn = 0;
            static_for<16>([&](auto cc) { constexpr int c = decltype(cc)::value; xn += dv[c] * dpp_row_bcast<c>(z); });
            if (lane >= l0 && lane < l0 + 16) { z = xn; xk[(kb & 1) * 16 + r] = xn; }
        }
        __syncthreads();
        if (i >= c0 + 16 && i < n) {
            T acc = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc += lrow[c] * ((c0 + c < n) ? xk[(kb & 1) * 16 + c] : T(0));
            z -= acc;
        }
    }
    // ---- backward: L^T x = z
    for (int kb = nb - 1; kb >= 0; --kb) {
        const int c0 = 16 * kb;
        T lcol[16];                                             // L[c0 .. c0 + 15][i], needed when i < c0
#pragma unroll
        for (int c = 0; c < 16; ++c) lcol[c] = lnext[c];
        if (kb > 0) load_col(kb - 1);
        if (wave == (c0 >> 6)) {
            const int l0 = c0 & 63, r = lane & 15;
            T dv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) dv[c] = Dinv[kb * 272 + c + 17 * r];                         // (inv L_kk)^T (r, c) = inv(c, r)
            T xn = 0;
            static_for<16>([&](auto cc) { constexpr int c = decltype(cc)::value; xn += dv[c] * dpp_row_bcast<c>(z); });
            if (lane >= l0 && lane < l0 + 16) { z = xn; xk[(kb & 1) * 16 + r] = xn; }
        }
        __syncthreads();
        if (i < c0) {
            T acc = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc += lcol[c] * ((c0 + c < n) ? xk[(kb & 1) * 16 + c] : T(0));
            z -= acc;
        }
    }
    if (i < n) xv[i] = z;
    __syncthreads();
}

// ---------------------------------------------------------------- ?posvx('E','L'), nrhs = 1, generic path (128 < n <= 256)
// A: n x n full symmetric, leading dimension lda (overwritten by its equilibrated form).
// b: right-hand side (overwritten by the scaled rhs). x: solution. s,r,w: n-vectors.
// F/ldf: factor storage (global, L2 resident). Returns info (0 = ok, k > 0 = leading minor k not
// positive definite). Collective over the workgroup. (n <= 128 runs posvx_lds, solve_lds.h.)
template <typename T, int NB>
__device__ __forceinline__ int posvx_device(int n, T* A, int lda, T* F, int ldf, T* s, T* b, T* x, T* r, T* w, T* red, long long* dbg = nullptr)
{
    static_assert(NB == 0, "the LDS path is posvx_lds");
    MIRLSQ_STAMP(dbg, 2);
    // inverse diagonal blocks, the current diagonal block and its reciprocal pivots
    __shared__ T gDinv[16 * 272];
    __shared__ T gblk[16 * 17];
    __shared__ T gpanel[kSolveThreads * 17];
    __shared__ T grd[16];
    __shared__ int ginfo[1];
    const int tid = threadIdx.x;
    const T eps = Lim<T>::eps / 2;              // dlamch('Epsilon')
    const T safmin = Lim<T>::min_normal;        // dlamch('Safe minimum')

    // ?poequ
    T di = tid < n ? A[tid + (size_t)tid * lda] : T(0);
    const T smin = block_min(tid < n ? di : Lim<T>::inf(), red);
    const T amax = block_max(tid < n ? di : -Lim<T>::inf(), red);
    bool rcequ = false;
    if (smin > 0) {
        const T scond = dsqrt(smin) / dsqrt(amax);
        if (tid < n) s[tid] = T(1) / dsqrt(di);
        // ?laqsy
        const T small = safmin / Lim<T>::eps, large = T(1) / small;
        rcequ = !(scond >= T(0.1) && amax >= small && amax <= large);
    }
    __syncthreads();
    if (rcequ) {
        // column j handled by the threads of one quarter-wave stride: no integer division
        for (int j = tid >> 6; j < n; j += kSolveThreads / kWave) {
            const T cj = s[j];
#pragma unroll 4
            for (int i = tid & 63; i < n; i += kWave) A[i + (size_t)j * lda] = cj * s[i] * A[i + (size_t)j * lda];
        }
        if (tid < n) b[tid] = s[tid] * b[tid];
    }
    __syncthreads();

    MIRLSQ_STAMP(dbg, 3);
    // ?lacpy + ?potrf 'L'
    {
        // factor in global memory, left-looking panels (potrf_panel) + the blocked triangular solves
        const int info = potrf_panel<T>(n, A, lda, F, ldf, gblk, grd, ginfo, gpanel, dbg);
        if (info != 0) return info;
        invert_diag_blocks_generic<T>(n, F, ldf, gDinv);
    }

    MIRLSQ_STAMP(dbg, 4);
    // ?potrs
    if (tid < n) x[tid] = b[tid];
    potrs_rows<T>(n, F, ldf, gDinv, x, gblk);

    MIRLSQ_STAMP(dbg, 5);
    // ?porfs: iterative refinement, ITMAX = 5
    const T safe1 = T(n + 1) * safmin, safe2 = safe1 / eps;
    T lstres = 3;
    for (int count = 1;; ++count) {
        // r = b - A x ; w = |b| + |A| |x|      (two threads per row, 16 loads in flight per thread)
        if (n % 2 == 0 && lda % 2 == 0) {
            // row PAIRS with 16-byte loads (A is column-major: rows 2 t, 2 t + 1 of a column are adjacent): twice the bytes per
            // round trip to L2 -- one workgroup reads the whole n x n matrix here, 28 us at n = 256 with 8-byte loads. Thread
            // (t = tid >> 1, h = tid & 1) sums the columns of half h for rows 2 t, 2 t + 1 in the same order as below: same bits.
            typedef T v2 __attribute__((ext_vector_type(2)));
            const int t = tid >> 1, h = tid & 1;
            const int i0 = 2 * t < n ? 2 * t : n - 2;
            const int k0 = h ? n / 2 : 0, k1 = h ? n : n / 2;
            const T* Ap = A;
            const T* xp = x;
            T ra = 0, rb = 0, wa = 0, wb = 0;
            // x through LDS (gpanel is free here), A in batches of 16 columns with TWO batches in flight beyond the one being
            // summed (always 16 loads a batch, indices clamped: a count the compiler can keep in vmcnt): the sweep pays the
            // latency of L2 once, not once per batch (19 us per residual at n = 256 with one batch at a time)
            const lds_ptr<T> xs = as_lds(gpanel);
            __syncthreads();
            if (tid < n) xs[tid] = xp[tid];
            __syncthreads();
            constexpr int kAheadR = 2, kRingR = kAheadR + 1;
            v2 av[kRingR][16];
            const int nbat = (k1 - k0 + 15) / 16;
            auto issue = [&](int bt, auto B) {
                constexpr int bb = decltype(B)::value;
                const int kb = k0 + 16 * (bt < nbat ? bt : nbat - 1);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = kb + u < k1 ? kb + u : k1 - 1;
                    av[bb][u] = *reinterpret_cast<const v2*>(Ap + i0 + (size_t)k * lda);
                }
            };
            static_for<kAheadR>([&](auto U) { issue(decltype(U)::value, U); });
            for (int b0 = 0; b0 < nbat; b0 += kRingR) {
                static_for<kRingR>([&](auto U) {
                    constexpr int u0 = decltype(U)::value;
                    const int bt = b0 + u0;
                    if (bt < nbat) {
                        issue(bt + kAheadR, IntC<(u0 + kAheadR) % kRingR>{});
                        const int kb = k0 + 16 * bt;
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            if (kb + u < k1) {
                                const T xu = xs[kb + u];
                                ra -= av[u0][u].x * xu; wa += dabs(av[u0][u].x) * dabs(xu);
                                rb -= av[u0][u].y * xu; wb += dabs(av[u0][u].y) * dabs(xu);
                            }
                    }
                });
            }
            ra += wave_shfl_xor(ra, 1); wa += wave_shfl_xor(wa, 1);
            rb += wave_shfl_xor(rb, 1); wb += wave_shfl_xor(wb, 1);
            if (2 * t < n && h == 0) {
                r[2 * t] = b[2 * t] + ra; w[2 * t] = dabs(b[2 * t]) + wa;
                r[2 * t + 1] = b[2 * t + 1] + rb; w[2 * t + 1] = dabs(b[2 * t + 1]) + wb;
            }
        } else
        for (int base = 0; base < n; base += kSolveThreads / 2) {
            const int i = base + (tid >> 1), h = tid & 1;
            const int ic = i < n ? i : n - 1;
            const int k0 = h ? n / 2 : 0, k1 = h ? n : n / 2;
            // plain pointers: A was written and x is rewritten between sweeps by other threads of this workgroup, so the
            // compiler must not treat these loads as invariant (no scalar-cache loads, no hoisting across the barriers)
            const T* Ap = A;
            const T* xp = x;
            T ri = 0, wi = 0;
            for (int kb = k0; kb < k1; kb += 16) {
                T av[16], xv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = kb + u < k1 ? kb + u : k1 - 1;
                    av[u] = Ap[ic + (size_t)k * lda];
                    xv[u] = xp[k];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (kb + u < k1) { ri -= av[u] * xv[u]; wi += dabs(av[u]) * dabs(xv[u]); }
            }
            ri += wave_shfl_xor(ri, 1);
            wi += wave_shfl_xor(wi, 1);
            if (i < n && h == 0) { r[i] = b[i] + ri; w[i] = dabs(b[i]) + wi; }
        }
        __syncthreads();
        if (count == 1) MIRLSQ_STAMP(dbg, 11);
        T qv = 0;
        if (tid < n) qv = (w[tid] > safe2) ? dabs(r[tid]) / w[tid] : (dabs(r[tid]) + safe1) / (w[tid] + safe1);
        const T berr = block_max(qv, red);
        if (count == 1) MIRLSQ_STAMP(dbg, 12);
        if (berr > eps && 2 * berr <= lstres && count <= 5) {
            potrs_rows<T>(n, F, ldf, gDinv, r, gblk);
            if (tid < n) x[tid] += r[tid];
            lstres = berr;
            __syncthreads();
            continue;
        }
        break;
    }
    if (rcequ && tid < n) x[tid] = s[tid] * x[tid];
    __syncthreads();
    MIRLSQ_STAMP(dbg, 6);
    return 0;
}

// ---------------------------------------------------------------- solveBoxQP, boxcqp.d:122-379
// Pm: n x n full symmetric (unscaled). q, l, u: n-vectors. x: in/out (holds the unconstrained
// solution on entry when skip_unconstrained). Returns BoxQPStatus; *iters = active-set passes.
// BOUNDED = false (k_lm_solve for problems whose lower / upper are all infinite, so qpl = -inf, qpu = +inf): the
// active-set loop is compiled out. It could only be entered with a NaN in the unconstrained solution (QP:216-219), where
// the reference's loop classifies every variable as free and leaves with `s == n` (QP:265, quirk Q8) and a status other
// than `solved` -- which is all the LM loop looks at (LS:1080): this variant returns numericError there. One inlined copy
// of posvx instead of two roughly halves the kernel and relieves its register allocation (512 VGPRs, 786 spilled SGPRs).
// NB > 0 (n <= 128): every system is solved by posvx_lds (solve_lds.h), which loads its matrix from global memory into
// LDS itself: the first one from `src0` (n x n full symmetric, leading dimension n) with `shift0` added on the diagonal
// -- J^T J and lambda for k_lm_solve, P and 0 for the standalone entry --, the reduced ones from sc.A. F is then the LDS
// block (LdsSolveCfg<NB>::ELEMS elements). NB == 0: the generic path with the factor in global memory (posvx_device).
template <typename T, int NB, bool BOUNDED = true>
__device__ __forceinline__ int box_qp_device(int n, const T* Pm, const T* q, const T* l, const T* u, T* x,
                             bool unconstrainedSolution, T relTol, T absTol, uint32_t maxIterations,
                             SolveScratch<T>& sc, T* F, int ldf, T* red, int* ired, int* iters, bool a_prefilled,
                             const T* src0, T shift0, bool* a_in_lds, LdsPreload<T, (NB > 0 ? NB : 1)>& pre0, bool preloaded0)
{
    const int tid = threadIdx.x;
    T* s = sc.vec;
    T* b = sc.vec + 1 * (size_t)n;
    T* r = sc.vec + 2 * (size_t)n;
    T* w = sc.vec + 3 * (size_t)n;
    T* la = sc.vec + 4 * (size_t)n;
    T* mu = sc.vec + 5 * (size_t)n;
    T* sX = sc.vec + 6 * (size_t)n;
    int32_t* SI = sc.ivec;
    int32_t* flags = sc.ivec + n;
    *iters = 0;
    if (n == 0) return 0;                                           // QP:162-163

    if constexpr (NB > 0) {
        if (!unconstrainedSolution) {                               // QP:168-214
            T xi = 0;
            bool scaled = false;
            const int info = posvx_lds<T, NB>(n, src0 ? src0 : Pm, n, shift0, tid < n ? -q[tid] : T(0), xi, F, red, ired + 8, sc.dbg, &scaled, pre0, preloaded0);
            if (a_in_lds) *a_in_lds = !scaled;                      // the LDS copy of A is src0 + shift0 I, unscaled
            if (info != 0) return 1;                                // QP:212-213 (info == n+1 is never produced)
            if (tid < n) x[tid] = xi;
            __syncthreads();
        }
    } else
    if (!unconstrainedSolution) {                                   // QP:168-214
        if (!a_prefilled) {                                            // QP:186-189
            const T* __restrict__ src = Pm;
            T* __restrict__ da = sc.A;
            const int nn = n * n;
            for (int base = tid; base < nn; base += 16 * kSolveThreads) {
                T v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int idx = base + u * kSolveThreads; v[u] = src[idx < nn ? idx : nn - 1]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int idx = base + u * kSolveThreads; if (idx < nn) da[idx] = v[u]; }
            }
        }
        if (tid < n) b[tid] = -q[tid];                              // QP:191
        __syncthreads();
        const int info = posvx_device<T, NB>(n, sc.A, n, F, ldf, s, b, x, r, w, red, sc.dbg);
        if (info != 0) return 1;                                    // QP:212-213 (info == n+1 is never produced)
    }

    {                                                               // QP:216-219
        int bad = 0;
        if (tid < n) bad = !(l[tid] <= x[tid] && x[tid] <= u[tid]);
        if (!block_or(bad, ired)) return 0;
    }
    if constexpr (!BOUNDED) return 1;
    if (a_in_lds) *a_in_lds = false;                                // the reduced systems reuse the LDS blocks

    if (!maxIterations) maxIterations = (uint32_t)n * 10 + 100;     // QP:224-226
    if (tid < n) { la[tid] = 0; mu[tid] = 0; }                      // QP:228-232
    __syncthreads();

    for (uint32_t step = 0; step < maxIterations; ++step) {         // QP:234
        *iters = (int)step + 1;
        // classification, QP:239-263 (element-parallel; the free-set list SI is built by a
        // ballot prefix sum, which preserves the reference's ascending order)
        int fl = 2;   // 2 = not an element
        if (tid < n) {
            const T xi = x[tid], li = l[tid], ui = u[tid];
            const T xl = xi - li, ux = ui - xi;
            if (xl < 0 || (xl < relTol + absTol * dabs(li) && la[tid] >= 0)) { fl = -1; x[tid] = li; mu[tid] = 0; }
            else if (ux < 0 || (ux < relTol + absTol * dabs(ui) && mu[tid] >= 0)) { fl = 1; x[tid] = ui; la[tid] = 0; }
            else { fl = 0; mu[tid] = 0; la[tid] = 0; }
            flags[tid] = fl;
        }
        const unsigned long long bal = __ballot(fl == 0);
        const int wave = tid >> 6, lane = tid & 63;
        __syncthreads();
        if (lane == 0) ired[1 + wave] = __popcll(bal);
        __syncthreads();
        int base = 0;
        for (int wv = 0; wv < wave; ++wv) base += ired[1 + wv];
        int sN = 0;
        for (int wv = 0; wv < kSolveThreads / kWave; ++wv) sN += ired[1 + wv];
        if (fl == 0) SI[base + __popcll(bal & ((1ull << lane) - 1))] = tid;
        __syncthreads();
        if (sN == n) break;                                         // QP:265-266 (quirk Q8)

        // reduced system, QP:282-305: thread ii assembles row ii of A_s and b_ii with a
        // Kahan-Babuska-Neumaier sum over the bound variables, j ascending as in the reference
        if (tid < sN) {
            const int i = SI[tid];
            T ks = q[i], kc = 0;
            int jj = 0;
            for (int j = 0; j < n; ++j) {
                const T pij = Pm[(size_t)j * n + i];
                const int fj = flags[j];
                if (fj) {
                    const T v = pij * (fj < 0 ? l[j] : u[j]);
                    const T t = ks + v;
                    if (dabs(ks) >= dabs(v)) kc += (ks - t) + v; else kc += (v - t) + ks;
                    ks = t;
                } else {
                    sc.A[(size_t)jj * sN + tid] = pij;              // A_s(tid, jj) (symmetric)
                    ++jj;
                }
            }
            b[tid] = -(ks + kc);
        }
        __syncthreads();
        if constexpr (NB > 0) {
            if (sN) {                                               // QP:307-325
                T xi = 0;
                const int info = posvx_lds<T, NB>(sN, sc.A, sN, T(0), tid < sN ? b[tid] : T(0), xi, F, red, ired + 8, nullptr, nullptr, pre0, false);
                if (info != 0) return 1;
                if (tid < sN) sX[tid] = xi;
            }
        } else if (sN) {                                            // QP:307-325
            const int info = posvx_device<T, NB>(sN, sc.A, sN, F, ldf, s, b, sX, r, w, red);
            if (info != 0) return 1;
        }
        if (tid < sN) x[SI[tid]] = sX[tid];                         // QP:327-329
        __syncthreads();

        // multipliers, QP:333-337
        if (tid < n && flags[tid]) {
            T v1 = 0, v2 = 0;
            for (int j = 0; j < tid; ++j) v1 += Pm[(size_t)j * n + tid] * x[j];
            for (int j = tid; j < n; ++j) v2 += Pm[(size_t)j * n + tid] * x[j];
            const T val = v1 + v2 + q[tid];
            if (flags[tid] < 0) la[tid] = val; else mu[tid] = -val;
        }
        __syncthreads();
        int again = 0;                                              // QP:339-347
        if (tid < n) {
            const int fi = flags[tid];
            if (fi < 0) again = !(la[tid] >= 0);
            else if (fi > 0) again = !(mu[tid] >= 0);
            else again = !(x[tid] >= l[tid] && x[tid] <= u[tid]);
        }
        if (block_or(again, ired)) continue;

        if (tid < n) x[tid] = dfmax(dfmin(x[tid], u[tid]), l[tid]); // QP:349 applyBounds
        __syncthreads();
        return 0;
    }
    return 2;                                                       // QP:378
}

// ---------------------------------------------------------------- one LM pass, n x n part (LmSolveArgs: solve_types.h)
// The body of k_lm_solve as a device function: ladder entry kc of `a`, collective over a workgroup of kSolveThreads threads;
// smem_raw: LdsSolveCfg<NB>::ELEMS elements of LDS when NB > 0. Also called by the resident-J cooperative solver
// (resident_kernel.h), whose workgroup 0 runs the n x n part of every pass inside the one launch.
// `pre` / preissued: the fused round's kernel has the values of J^T J in registers already (loaded at ITS entry, the Broyden pass's
// rank-two term added on the way -- lm_round_head); otherwise the body issues the loads itself.
// `fin` (NB = 0 only, the fused round above n = 128): the Broyden pass's rank-two term has NOT been applied to J^T J yet -- the
// copy loop below reads the matrix anyway and adds it on the way (v, dx staged in LDS by lm_round_head), writing J^T J back.
template <typename T> struct FinTerm { const T* v; const T* dxs; T uu; T* JJw; bool on; };
template <typename T, int NB, bool BOUNDED = true>
__device__ __forceinline__ void lm_solve_body(const LmSolveArgs<T>& a, const int kc, unsigned char* smem_raw,
                                              LdsPreload<T, (NB > 0 ? NB : 1)>& pre, const bool preissued,
                                              const FinTerm<T> fin = FinTerm<T>{nullptr, nullptr, T(0), nullptr, false})
{
    __shared__ T red[8];
    __shared__ int ired[12];
    const int n = a.n, tid = threadIdx.x;
    const int ldf = n | 1;
    SolveScratch<T> sc = a.sc[kc];
    T* dx_out = a.dx + (size_t)kc * n;
    T* trial_out = a.trial + (size_t)kc * n;
    T* F;
    if constexpr (NB > 0) F = reinterpret_cast<T*>(smem_raw); else F = sc.Fg;
    T* qpl = sc.vec + 7 * (size_t)n;
    T* qpu = sc.vec + 8 * (size_t)n;
    T* xq = sc.vec + 10 * (size_t)n;

    MIRLSQ_STAMP(sc.dbg, 0);
    if (sc.dbg && threadIdx.x == 0) sc.dbg[9] = clock64();
    // the LDS path: the loads of J^T J go out now and are collected inside ?posvx, behind this prologue (solve_lds.h)
    if constexpr (NB > 0) { if (!preissued) lds_load_issue<T, NB>(n, a.JJ, n, pre); }
    T jy_inf = 0;
    if (a.check_grad) {
        jy_inf = block_max(tid < n ? dabs(a.Jy[tid]) : T(0), red);       // |Jy[iamax(Jy)]|, LS:1053
        __syncthreads();
    }
    if (a.check_grad && tid == 0 && kc == 0) a.st->jy_inf = jy_inf;
    // gradient test, LS:1053: stop before touching lambda when ||Jy||_inf <= gradTolerance
    if (a.check_grad && !(jy_inf > a.set.gradTolerance)) {
        if (tid == 0) { ChainRec<T> r{}; r.flags = kFlagGradSmall; a.rec[kc] = r; }
        // (fin.on: the copy loop below, which adds the Broyden pass's rank-two term to J^T J on its way, is not reached -- but the
        // pass HAS happened, and after a failed gradient test with an aged Jacobian the next pass solves with this J^T J again
        // (LS:1053-1062, quirk Q4): the term goes into memory here, entry by entry k_lr_finish's expression)
        if (fin.on) {
            for (int idx = tid; idx < n * n; idx += kSolveThreads) {
                const int i = idx / n, j = idx - i * n;
                const bool lo = i >= j;
                fin.JJw[idx] += lr_jj_term(fin.v[lo ? i : j], fin.v[lo ? j : i], fin.dxs[lo ? i : j], fin.dxs[lo ? j : i], fin.uu);
            }
        }
        return;
    }

    // lambda_0, LS:1067-1072 (first element of maximum |diag|, as i?amax picks it)
    T lambda = (kc == 0 && (a.lambda_from_state || a.lambda_from_device)) ? a.st->lambda : a.lam[kc];
    if (kc == 0 && a.lambda_from_state && !(lambda >= a.set.minLambda)) {
        const T dg = tid < n ? dabs(a.JJ[(size_t)tid * n + tid]) : T(-1);
        const T mx = block_max(dg, red);
        int cand = (tid < n && dg == mx) ? tid : 0x7fffffff;
#pragma unroll
        for (int k = 1; k < kWave; k <<= 1) { const int o = __shfl_xor(cand, k, kWave); cand = o < cand ? o : cand; }
        __syncthreads();
        if ((tid & 63) == 0) ired[4 + (tid >> 6)] = cand;
        __syncthreads();
        int first = ired[4];
        for (int wv = 1; wv < kSolveThreads / kWave; ++wv) first = ired[4 + wv] < first ? ired[4 + wv] : first;
        lambda = T(0.001) * a.JJ[(size_t)first * n + first];
        if (!(lambda >= a.set.minLambda)) lambda = 1;
    }

    // step bounds LS:1074-1077, P = JJ + lambda I LS:1078-1079 (JJ itself is never modified, so
    // the save/restore of its diagonal at LS:1078/1094 is not needed)
    if (tid < n) { qpl[tid] = a.lower[tid] - a.x[tid]; qpu[tid] = a.upper[tid] - a.x[tid]; }
    if constexpr (NB == 0 || BOUNDED) {
        // Pm = A = JJ + lambda I in global memory: the generic path factors A, the BOXCQP loop reads Pm. (The LDS path of
        // an unbounded problem loads JJ + lambda I straight into LDS and needs neither.) 16 loads in flight per thread (a
        // plain copy loop serialises on may-alias load/store ordering)
        const T* __restrict__ src = a.JJ;
        T* __restrict__ dp = sc.Pm;
        T* __restrict__ da = sc.A;
        const int nn = n * n;
        // Pm is what the BOXCQP loop reads: an unbounded problem (BOUNDED = false) never gets there
        constexpr bool kNeedPm = BOUNDED;
        if (nn % 2 == 0) {
            // 16-byte accesses, 16 in flight per thread: one workgroup has to move 3 n^2 elements here (60 us at n = 256 with
            // 8-byte accesses and both copies)
            typedef T v2 __attribute__((ext_vector_type(2)));
            const v2* __restrict__ s2 = reinterpret_cast<const v2*>(src);
            v2* __restrict__ p2 = reinterpret_cast<v2*>(dp);
            v2* __restrict__ a2 = reinterpret_cast<v2*>(da);
            v2* __restrict__ w2 = reinterpret_cast<v2*>(fin.JJw);
            const int np = nn / 2;
            for (int base = tid; base < np; base += 16 * kSolveThreads) {
                v2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int idx = base + u * kSolveThreads; v[u] = s2[idx < np ? idx : np - 1]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int idx = base + u * kSolveThreads;
                    if (idx < np) {
                        v2 t = v[u];
                        const int i = (2 * idx) / n, j = 2 * idx - i * n;      // n even: both elements in row i
                        if (fin.on) {
                            const T vi = fin.v[i], di = fin.dxs[i];
                            const T vj0 = fin.v[j], dj0 = fin.dxs[j], vj1 = fin.v[j + 1], dj1 = fin.dxs[j + 1];
                            const bool l0 = i >= j, l1 = i >= j + 1;
                            t.x += lr_jj_term(l0 ? vi : vj0, l0 ? vj0 : vi, l0 ? di : dj0, l0 ? dj0 : di, fin.uu);
                            t.y += lr_jj_term(l1 ? vi : vj1, l1 ? vj1 : vi, l1 ? di : dj1, l1 ? dj1 : di, fin.uu);
                            w2[idx] = t;
                        }
                        if (j == i) t.x += lambda;
                        if (j + 1 == i) t.y += lambda;
                        if constexpr (kNeedPm) p2[idx] = t;
                        a2[idx] = t;
                    }
                }
            }
        } else {
            for (int base = tid; base < nn; base += 16 * kSolveThreads) {
                T v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int idx = base + u * kSolveThreads; v[u] = src[idx < nn ? idx : nn - 1]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int idx = base + u * kSolveThreads;
                    if (idx < nn) {
                        T t = v[u];
                        const int i = idx / n, j = idx - i * n;
                        if (fin.on) {
                            const bool lo = i >= j;
                            t += lr_jj_term(fin.v[lo ? i : j], fin.v[lo ? j : i], fin.dxs[lo ? i : j], fin.dxs[lo ? j : i], fin.uu);
                            fin.JJw[idx] = t;
                        }
                        if (i == j) t += lambda;
                        if constexpr (kNeedPm) dp[idx] = t;
                        da[idx] = t;
                    }
                }
            }
        }
    }
    __syncthreads();

    MIRLSQ_STAMP(sc.dbg, 1);
    int qp_iters = 0;
    bool a_in_lds = false;
    const int qp = box_qp_device<T, NB, BOUNDED>(n, sc.Pm, a.Jy, qpl, qpu, xq, false, a.set.qpRelTolerance, a.set.qpAbsTolerance,
                                          a.set.qpMaxIterations, sc, F, ldf, red, ired, &qp_iters, true, a.JJ, lambda, &a_in_lds,
                                          pre, NB > 0);   // LS:1080

    MIRLSQ_STAMP(sc.dbg, 7);
    int flags = 0;
    T ndd = 0, pred = 0, xn = 0;
    if (qp == 0) {
        T d = 0, tr = 0;
        int moved = 0;
        if (tid < n) {
            d = xq[tid];
            if (!(d <= d)) flags = kFlagDxNaN;                       // LS:1087
            const T xi = a.x[tid];
            d = d + xi;                                              // LS:1096
            d = d - xi;                                              // LS:1097
            dx_out[tid] = d;
            tr = dfmax(dfmin(d + xi, a.upper[tid]), a.lower[tid]);   // LS:1108-1110
            trial_out[tid] = tr;
            if (!(tr <= tr)) flags |= kFlagXNaN;
            moved = !(tr == xi);                                     // NaN counts as moved
        }
        // predicted reduction with the UNDAMPED JJ, LS:1141-1142: t = JJ dx + 2 Jy ; pred = -(t . dx)
        T ti = 0;
        if constexpr (NB > 0) {
            // The step is staged in LDS (the solve's vectors are free now). When the LDS blocks still hold the unscaled
            // A = JJ + lambda I of this pass (no equilibration, no active-set iteration), JJ dx is read from them: every
            // off-diagonal entry of A IS the entry of JJ, and the diagonal term uses JJ_ii fetched from memory -- the same
            // products as the sweep over global memory below, without its 64 dependent L2 round trips per thread.
            using LC = LdsSolveCfg<NB>;
            T* stage = F + LC::XV_OFF;
            const int i = tid >> 1, h = tid & 1;
            const T jdiag = (a_in_lds && i < n) ? a.JJ[(size_t)i * n + i] : T(0);
            if (tid < LC::NV) stage[tid] = tid < n ? d : T(0);
            __syncthreads();
            if (a_in_lds) {
                const T* Ab = F + LC::A_OFF;
                const int nbl = (n + 15) >> 4, I = i >> 4, r = i & 15;
                if (i < n) {
                    T acc[4] = {0, 0, 0, 0};
                    for (int J = h; J < nbl; J += 2) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            T av = J <= I ? Ab[blk_off(I, J, r, c)] : Ab[blk_off(J, I, c, r)];
                            av = (16 * J + c == i) ? jdiag : av;
                            acc[c & 3] += av * stage[16 * J + c];
                        }
                    }
                    ti = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                    if (h == 0) ti = ti + 2 * a.Jy[i];
                    ti = ti * stage[i];
                }
            } else if (i < n) {
                const int j0 = h ? n / 2 : 0, j1 = h ? n : n / 2;
                const T* __restrict__ jj = a.JJ;
#pragma unroll 16
                for (int j = j0; j < j1; ++j) ti += jj[(size_t)j * n + i] * stage[j];
                if (h == 0) ti = ti + 2 * a.Jy[i];
                ti = ti * stage[i];
            }
        } else {
            // thread pair (i, h) sums half of row i (n <= 128) or thread i sums the whole row
            const bool pair = n <= kSolveThreads / 2;
            const int i = pair ? tid >> 1 : tid, h = pair ? tid & 1 : 0;
            if (i < n) {
                const int j0 = pair && h ? n / 2 : 0, j1 = pair && !h ? n / 2 : n;
#pragma unroll 16
                for (int j = j0; j < j1; ++j) ti += a.JJ[(size_t)j * n + i] * dx_out[j];
                if (h == 0) ti = ti + 2 * a.Jy[i];
                ti = ti * dx_out[i];
            }
        }
        // one fused workgroup reduction: ||dx||^2 (LS:1099), t . dx, max |trial|, the OR of the flags and "some entry moved"
        {
            __shared__ T fr[3][kSolveThreads / kWave];
            __shared__ int fi[kSolveThreads / kWave];
            const T s0 = wave_sum(d * d), s1 = wave_sum(ti), m0 = wave_max(dabs(tr));
            const unsigned long long bnan = __ballot(flags & kFlagDxNaN), bx = __ballot(flags & kFlagXNaN), bm = __ballot(moved);
            const int wv = tid >> 6;
            __syncthreads();
            if ((tid & 63) == 0) {
                fr[0][wv] = s0; fr[1][wv] = s1; fr[2][wv] = m0;
                fi[wv] = (bnan ? kFlagDxNaN : 0) | (bx ? kFlagXNaN : 0) | (bm ? 0x10000 : 0);
            }
            __syncthreads();
            ndd = (fr[0][0] + fr[0][1]) + (fr[0][2] + fr[0][3]);
            pred = -((fr[1][0] + fr[1][1]) + (fr[1][2] + fr[1][3]));
            T amx = fr[2][0];
#pragma unroll
            for (int w2 = 1; w2 < kSolveThreads / kWave; ++w2) amx = fr[2][w2] > amx ? fr[2][w2] : amx;
            const int fo = fi[0] | fi[1] | fi[2] | fi[3];
            flags = fo & 0xffff;
            if (!(fo & 0x10000)) flags |= kFlagNullStep;
            // ||trial||_2 for the relTolerance test, LS:1164 (scaled like ?nrm2)
            T sc2 = 0;
            if (tid < n && amx > 0) { const T v = tr / amx; sc2 = v * v; }
            xn = amx > 0 ? amx * dsqrt(block_sum(sc2, red)) : T(0);
        }
        if (!(dsqrt(ndd) < a.set.maxStep)) flags |= kFlagStepTooLong; // LS:1101
    }
    MIRLSQ_STAMP(sc.dbg, 8);
    if (sc.dbg && threadIdx.x == 0) sc.dbg[10] = clock64();
    if (tid == 0) {
        ChainRec<T> r{};
        r.lambda = lambda; r.new_dx_dot = ndd; r.predicted = pred; r.trial_xnorm = xn;
        r.qp_status = qp; r.qp_iterations = qp_iters; r.flags = flags;
        a.rec[kc] = r;
    }
}

template <typename T, int NB, bool BOUNDED = true>
__device__ __forceinline__ void lm_solve_body(const LmSolveArgs<T>& a, const int kc, unsigned char* smem_raw)
{
    LdsPreload<T, (NB > 0 ? NB : 1)> pre;
    lm_solve_body<T, NB, BOUNDED>(a, kc, smem_raw, pre, false);
}

// Head of a FUSED round (a.fused; LmSolveArgs): the decision of the previous round's trial (LS:1112-1161; its sum of squares came
// in with the all-reduced sweep vector), published to the host at once, and -- if a Broyden pass follows -- that pass's n x n
// side (LS:1003-1006, 1052, 1065 as k_lr_finish forms them). false: the kernel is done (no pass follows, the host takes over).
// fin_v: 2 n elements of LDS scratch (the LDS solve's dynamic region is free until the body starts: at n = 128 it leaves 4 KB of a
// CU's 160 KB for every static variable of the kernel).
// NB > 0 (n <= 128, the LDS solve): `pre` holds J^T J, loaded at kernel entry so that the memory latency runs behind the
// decision; the pass's rank-two term is added IN THESE REGISTERS (entry by entry k_lr_finish's expression: the same bits), the
// updated matrix goes back to memory (both triangles) for the passes to come, and the body commits the registers to LDS --
// one trip through memory where a separate finish costs a read-modify-write AND the body's read (7.5 us at n = 128 -> 1.5).
template <typename T, int NB, bool BOUNDED>
__device__ __forceinline__ bool lm_round_head(const LmSolveArgs<T>& a, T* fin_v, LdsPreload<T, (NB > 0 ? NB : 1)>& pre)
{
    MIRLSQ_STAMP(a.sc[0].dbg, 26);
    const int n = a.n, tid = threadIdx.x;
    // Everything the head reads from memory is requested NOW, before the decision is known (one trial: entry 0 of the ladder;
    // the sweep vector and the pending steps do not depend on the decision): ONE memory latency for the whole head instead of
    // a chain of five (decision -> point and step -> sweep vector -> pending steps -> J^T J).
    const bool el = tid < n;
    const int t0 = el ? tid : 0;
    DecidePre<T> dp;
    dp.xv = a.dec.trial[t0];
    dp.dv = a.dec.dx_chain[t0];
    const T* __restrict__ lr = a.fin_lr;
    const int k = a.fin_k;
    const T lv = lr[t0], lg = lr[n + t0];
    const T uu = lr[2 * n + 2 * kLrMax], uy = lr[2 * n + 2 * kLrMax + 1];
    T dl[kLrMax], wl[kLrMax], hl[kLrMax];
#pragma unroll
    for (int l = 0; l < kLrMax; ++l) {
        const bool on = l < k;
        dl[l] = a.fin_D[(size_t)(on ? l : 0) * n + t0];
        wl[l] = lr[2 * n + (on ? l : 0)];
        hl[l] = lr[2 * n + kLrMax + (on ? l : 0)];
    }
    if (!decide_chain_body(a.dec, &dp, a.sc[0].dbg)) return false;
    MIRLSQ_STAMP(a.sc[0].dbg, 27);
    // the vectors of k_lr_finish (its expressions, its order: the same bits): v = J_{k-1}^T u, J^T y, D_k = dx
    {
        T s = lv, g = lg;
#pragma unroll
        for (int l = 0; l < kLrMax; ++l) if (l < k) s = dfma(dl[l], wl[l], s);
#pragma unroll
        for (int l = 0; l < kLrMax; ++l) if (l < k) g = dfma(dl[l], hl[l], g);
        g = dfma(dp.dv, uy, g);
        if (el) {
            fin_v[tid] = s;
            fin_v[n + tid] = dp.dv;
            a.fin_Jy[tid] = g;
            a.fin_D[(size_t)k * n + tid] = dp.dv;
        }
    }
    lds_barrier();                                           // (fin_v; the stores to memory drain behind the arithmetic below)
    const T* v = fin_v;
    const T* dxs = fin_v + n;
    if constexpr (NB > 0) {
        T* __restrict__ JJ = a.fin_JJ;
        const int c = tid & 15, r = tid >> 4;
        // a thread's entries are (16 I + r, 16 J + c): 2 NB values of v and of the step cover them all -- read once, then the NB
        // (NB + 1) / 2 terms are register arithmetic (one LDS round trip per ENTRY was 2 of this phase's 4.4 us at n = 128)
        T vi[NB], di[NB], vj[NB], dj[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            const int gi = 16 * I + r, gj = 16 * I + c;
            vi[I] = v[gi < n ? gi : 0]; di[I] = dxs[gi < n ? gi : 0];
            vj[I] = v[gj < n ? gj : 0]; dj[I] = dxs[gj < n ? gj : 0];
        }
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J) {
                const int gi = 16 * I + r, gj = 16 * J + c;
                if (gi < n && gj < n) {
                    const bool up = gi < gj;                 // (only inside a diagonal block)
                    const T vh = up ? vj[J] : vi[I], vl = up ? vi[I] : vj[J], dh = up ? dj[J] : di[I], dl = up ? di[I] : dj[J];
                    const T t = pre.v[I * (I + 1) / 2 + J] + lr_jj_term(vh, vl, dh, dl, uu);
                    pre.v[I * (I + 1) / 2 + J] = t;
                    JJ[(size_t)gi * n + gj] = t;
                    if (I != J) JJ[(size_t)gj * n + gi] = t;
                }
            }
        if (el) pre.diag += lr_jj_term(v[tid], v[tid], dxs[tid], dxs[tid], uu);
    }
    // (NB = 0: the body's copy of J^T J adds the term on its way -- FinTerm; v and the step stay in fin_v)
    // fin_v is free (NB > 0). UNBOUNDED problems do not wait for the stores of J^T J, J^T y and D_k: a thread reads back what it wrote
    // itself (J^T y, program order), the matrix is read from memory again only behind the solve's own barriers (the
    // prediction), D_k by later kernels. A BOUNDED problem's body copies J^T J from memory into the matrices of the BOXCQP
    // loop at once -- other threads' entries: the stores must have landed (a full barrier; without it the copy raced with
    // them, tests/test_gpu_fused_rounds.py).
    if constexpr (NB > 0 && BOUNDED) __syncthreads();
    else if constexpr (NB > 0) lds_barrier();
    MIRLSQ_STAMP(a.sc[0].dbg, 28);
    return true;
}

template <typename T, int NB, bool BOUNDED = true>
__global__ __launch_bounds__(kSolveThreads) void k_lm_solve(LmSolveArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    LdsPreload<T, (NB > 0 ? NB : 1)> pre;
    FinTerm<T> fin{nullptr, nullptr, T(0), nullptr, false};
    if (a.fused) {
        T* fin_v;
        if constexpr (NB > 0) { fin_v = reinterpret_cast<T*>(smem_raw); lds_load_issue<T, NB>(a.n, a.JJ, a.n, pre); }
        else { __shared__ T fin_static[2 * kSolveMaxN]; fin_v = fin_static; }
        if (!lm_round_head<T, NB, BOUNDED>(a, fin_v, pre)) return;
        if constexpr (NB == 0) fin = FinTerm<T>{fin_v, fin_v + a.n, a.fin_lr[2 * a.n + 2 * kLrMax], a.fin_JJ, true};
    }
    lm_solve_body<T, NB, BOUNDED>(a, (int)blockIdx.x, smem_raw, pre, a.fused != 0 && NB > 0, fin);  // blockIdx.x: chain step
}

// n <= 16, f64: the same pass on ONE wave per ladder entry, every matrix a row per lane (solve_wave16.h): no LDS, no barrier,
// no global scratch -- 17 us -> see DESIGN.md for the launch-chain path of cfg 2 (n = 16). Compiled once, for 16 rows; rows
// n .. 15 are identity rows.
}  // namespace mirlsq
#include "solve_wave16.h"
namespace mirlsq {
template <bool BOUNDED>
__global__ __launch_bounds__(kWave) void k_lm_solve_wave(LmSolveArgs<double> a)
{
    const int n = a.n, kc = blockIdx.x, lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    __shared__ double fin_v[2 * kW16];
    if (a.fused) {
        LdsPreload<double, 1> none;
        if (!lm_round_head<double, 0, BOUNDED>(a, fin_v, none)) return;
    }
    const bool el = r < n;
    const int rc = el ? r : 0;
    double JJrow[kW16];
#pragma unroll
    for (int k = 0; k < kW16; ++k) { const double v = a.JJ[(size_t)rc * n + (k < n ? k : 0)]; JJrow[k] = (el && k < n) ? v : 0.0; }
    double djj_l = a.JJ[(size_t)rc * n + rc];
    if (a.fused) {
        // the Broyden pass's rank-two term (k_lr_finish's expression, entry by entry) on the row this lane holds; group 0 writes
        // the updated J^T J back for the passes to come
        const double uu = a.fin_lr[2 * n + 2 * kLrMax];
        const double vr = fin_v[rc], dr = fin_v[n + rc];
#pragma unroll
        for (int k = 0; k < kW16; ++k) {
            const int kk = k < n ? k : 0;
            const double vk = fin_v[kk], dk = fin_v[n + kk];
            const bool lower = rc >= kk;
            const double t = lr_jj_term(lower ? vr : vk, lower ? vk : vr, lower ? dr : dk, lower ? dk : dr, uu);
            if (el && k < n) {
                JJrow[k] += t;
                if (g == 0) a.fin_JJ[(size_t)rc * n + k] = JJrow[k];
            }
        }
        djj_l += lr_jj_term(vr, vr, dr, dr, uu);
    }
    const double jy_l = a.Jy[rc], x_l = a.x[rc], lo_l = a.lower[rc], up_l = a.upper[rc];
    const double djj = el ? djj_l : 0.0, Jy_r = el ? jy_l : 0.0, x_r = el ? x_l : 0.0;
    const double lo_r = el ? lo_l : -Lim<double>::inf(), up_r = el ? up_l : Lim<double>::inf();
    const double lambda = (kc == 0 && (a.lambda_from_state || a.lambda_from_device)) ? a.st->lambda : a.lam[kc];
    if (a.check_grad && kc == 0) {
        const double jy_inf = row16_max(fabs(Jy_r));
        if (lane == 0) a.st->jy_inf = jy_inf;
    }
    Wave16Level lv;
    double d, t;
    wave16_lm_solve<kW16, BOUNDED>(JJrow, djj, Jy_r, x_r, lo_r, up_r, lambda, 1.0, a.check_grad != 0, kc == 0 && a.lambda_from_state != 0,
                                   a.set, lv, d, t, n, true);
    if (g == 0 && el && !(lv.flags & kFlagGradSmall)) { a.dx[(size_t)kc * n + r] = d; a.trial[(size_t)kc * n + r] = t; }
    if (lane == 0) {
        ChainRec<double> rec{};
        rec.lambda = lv.lambda; rec.new_dx_dot = lv.ndd; rec.predicted = lv.pred; rec.trial_xnorm = lv.xnorm;
        rec.qp_status = lv.qp_status; rec.qp_iterations = lv.qp_iters; rec.flags = lv.flags;
        a.rec[kc] = rec;
    }
}

// standalone BOXCQP (mir_solve_box_qp_gpu_*; BoxQpArgs: solve_types.h)
template <typename T, int NB>
__global__ __launch_bounds__(kSolveThreads) void k_box_qp(BoxQpArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ T red[8];
    __shared__ int ired[12];
    const int n = a.n;
    T* F;
    if constexpr (NB > 0) F = reinterpret_cast<T*>(smem_raw); else F = a.sc.Fg;
    // symmetrise the lower triangle (QP:109: only the lower triangle of P is meaningful)
    for (int idx = threadIdx.x; idx < n * n; idx += kSolveThreads) {
        const int i = idx / n, j = idx % n;
        a.sc.Pm[idx] = i >= j ? a.P[(size_t)i * n + j] : a.P[(size_t)j * n + i];
    }
    __syncthreads();
    int it = 0;
    LdsPreload<T, (NB > 0 ? NB : 1)> pre;
    const int st = box_qp_device<T, NB>(n, a.sc.Pm, a.q, a.l, a.u, a.x, a.unconstrained != 0, a.relTol, a.absTol,
                                 a.maxIterations, a.sc, F, n | 1, red, ired, &it, false, nullptr, T(0), nullptr, pre, false);
    if (threadIdx.x == 0) { a.out[0] = st; a.out[1] = it; }
}

}  // namespace mirlsq
