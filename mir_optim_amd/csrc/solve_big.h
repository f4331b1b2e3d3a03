// solve_big.h -- the n x n part of an LM pass for ANY n (used above n = 256): one workgroup of 512 threads per
// lambda-ladder entry, matrices in global memory (L2 / Infinity-Cache resident), every n-vector loop strided.
//
// Same job and same references as solve_kernel.h:
//   least_squares.d:1053 (gradient test), 1067-1079 (lambda_0, P = J^T J + lambda I, step bounds), 1080 -> boxcqp.d:122-379
//   (solveBoxQP with ?posvx('E','L') = ?poequ + ?laqsy + ?potrf + ?potrs + ?porfs and the BOXCQP active-set loop),
//   1087-1110 (NaN guard, step rounding, trial point), 1141-1142 (predicted reduction), 1164 (norm of the trial point).
// The reference places no limit on n (least_squares.d:911-926 carves the workspace for any n); k_lm_solve keeps one
// element per thread and stops at 256, this file does not. It favours plain loops over tuning: the factorisation is
// the left-looking panel scheme of potrf_panel (MFMA 16x16x4 for the contribution of the earlier panels) walked over
// 512-row chunks, the triangular solves go block by block with one barrier per block. One CU does n^3 / 3 flops here,
// which is of the order of what the whole GPU spends on J^T J (m n^2) when m is a few hundred times n -- so from round 5 on
// the ladder entry's workgroup has HELPER workgroups in the same launch (solve_coop.h): they form the look-ahead blocks of
// ?potrf's update, share the n^2 matrix-vector sweeps and the copies of J^T J; everything serial stays here. ?potrs runs one
// or two rows per thread with the inverses of the diagonal blocks in LDS up to n = 1024.
#pragma once

#include "common.h"
#include "solve_kernel.h"
#include "solve_coop.h"

namespace mirlsq {

constexpr int kBigThreads = 512;
constexpr int kBigWaves = kBigThreads / kWave;
constexpr int kBigRowsMaxN = 2 * kBigThreads;  // ?potrs with one or two rows per thread and the diagonal blocks' inverses in LDS

// ---------------------------------------------------------------- workgroup collectives over strided partials
template <typename T, typename WaveOp, typename Op>
__device__ inline T big_reduce(T v, WaveOp wop, Op op, T* red /* >= kBigWaves */)
{
    v = wop(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = red[0];
#pragma unroll
    for (int w = 1; w < kBigWaves; ++w) r = op(r, red[w]);
    return r;
}
template <typename T> __device__ inline T big_sum(T v, T* red) { return big_reduce(v, [](T a) { return wave_sum(a); }, [](T a, T b) { return a + b; }, red); }
template <typename T> __device__ inline T big_max(T v, T* red) { return big_reduce(v, [](T a) { return wave_max(a); }, [](T a, T b) { return a > b ? a : b; }, red); }
template <typename T> __device__ inline T big_min(T v, T* red) { return big_reduce(v, [](T a) { return wave_min(a); }, [](T a, T b) { return a < b ? a : b; }, red); }
__device__ inline int big_or(int v, int* ired /* >= kBigWaves */)
{
    const unsigned long long b = __ballot(v != 0);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ired[threadIdx.x >> 6] = b ? 1 : 0;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < kBigWaves; ++w) r |= ired[w];
    return r;
}

template <typename T>
struct BigLds {                 // LDS of the big-n kernels (dynamic: 150 KB in double)
    T red[kBigWaves];           // (first: with span at LDS offset 0, hipcc 7.2 fails to select a null check of its generic address
    int ired[kBigWaves + 4];    //  in the float build -- "V_CMP_NE_U32 0, $src_shared_base")
    T span[kBigThreads * 17];   // S of the current chunk (row per thread after the MFMA stage)
    T dinv[(kBigThreads / 16) * 272];   // n <= kBigThreads: inverses of the 16 x 16 diagonal blocks of the factor (potrs_big_rows);
                                // kBigThreads < n <= kBigRowsMaxN: span AND dinv together hold the 64 inverses (span is free
                                // between factorisations) -- the two arrays must stay adjacent
    T blk[16 * 17];             // the current diagonal block L_kk
    T rd[16];                   // its reciprocal pivots
    T xk[32];                   // published solution block of a triangular-solve step (two slots: potrs_big_rows alternates)
    T xs[kBigRowsMaxN + 16];    // n <= kBigRowsMaxN: the vector of a matrix-vector sweep (above: span)
};
static_assert(offsetof(BigLds<double>, dinv) == offsetof(BigLds<double>, span) + sizeof(double) * kBigThreads * 17, "span and dinv adjacent");

// ---------------------------------------------------------------- ?potrf 'L', left-looking by 16-column panels
// A: n x n full symmetric (lda), F: the factor (ldf), both column-major in global memory. Collective; returns info.
template <typename T>
__device__ __noinline__ int potrf_big(int n, const T* A_, int lda, T* F_, int ldf, BigLds<T>& sm, long long* dbg = nullptr,
                                      CoopCtx* cc = nullptr, T* cS_ = nullptr)
{
    // helpers (solve_coop.h, kCoopPotrfT): while this workgroup factors the kCoopPanels (2) panels of a kCoopBlockCols (32)-column
    // block, the helpers form the next block's update from the columns that are final (blocks <= B - 2 for block B): step 1 below
    // then only covers the panels of blocks B - 1 and B (<= 2 kCoopPanels - 1 = 3 of them). The factor travels through agent-scope
    // stores; the helpers' part comes back through one of two n x kCoopBlockCols buffers at cS.
    const bool coop = cc && cc->W > 1 && cS_ && !cc->failed;
    bool pending = false;                                       // a kCoopPotrfT job is out
    // address spaces spelled out: through the generic pointers of an out-of-line function every access is a FLAT instruction, which
    // counts on lgkmcnt as well (16 outstanding at most, shared with the LDS reads) -- no ring of loads survives that
    const gbl_cptr<T> A = as_global(A_);
    const gbl_ptr<T> F = as_global_w(F_);
    const lds_ptr<T> span = as_lds(sm.span), blk = as_lds(sm.blk), rd = as_lds(sm.rd);
    long long ph[4] = {0, 0, 0, 0};       // DEBUG_SOLVE: time in steps 1, 2, 3 summed over the panels (thread 0)
    long long ph_job = 0;                 // ... of step 1: publishing / collecting the helpers' jobs
    using Acc = typename Mma<T>::Acc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int nblk = (n + 15) / 16;
    if (tid == 0) sm.ired[kBigWaves] = 0;
    __syncthreads();
    for (int k = 0; k < nblk; ++k) {
        const int c0 = 16 * k;
        const int B = k / kCoopPanels;                          // the look-ahead block of this panel
        if (coop && k % kCoopPanels == 0) {
            if (dbg && tid == 0) ph[3] = wall_clock64();
            if (pending) {
                coop_collect(*cc, kCoopPotrfT);
                pending = false;
                if (cc->failed) return n + 2;                   // (reported as a failed factorisation: numericError upstream)
            }
            if (B >= 1 && kCoopBlockCols * (B + 1) < n) {       // block B + 1: the columns of blocks <= B - 1 are final now
                const unsigned long long d[6] = {(unsigned long long)(uintptr_t)F_, (unsigned long long)(uintptr_t)(cS_ + (size_t)((B + 1) & 1) * n * kCoopBlockCols),
                                                 (unsigned long long)(unsigned)n | ((unsigned long long)(unsigned)ldf << 32), (unsigned long long)(B + 1), 0ull, 0ull};
                coop_publish(*cc, kCoopPotrfT | kCoopRelease, d);      // (the factor is stored with ordinary stores: written back here)
                pending = true;
            }
            if (dbg && tid == 0) { const long long t = wall_clock64(); ph[0] += t - ph[3]; ph_job += t - ph[3]; }
        }
        const int jstart = coop ? (B >= 1 ? kCoopPanels * (B - 1) : 0) : 0;     // step 1 covers panels jstart .. k - 1
        const bool useT = coop && B >= 2;                            // ... the helpers' buffer the columns before them
        const gbl_cptr<T> Tk = as_global(cS_ + (useT ? (size_t)(B & 1) * n * kCoopBlockCols + 16 * (k % kCoopPanels) : 0));
        for (int base = (c0 / kBigThreads) * kBigThreads; base < n; base += kBigThreads) {
            if (dbg && tid == 0) ph[3] = wall_clock64();
            // the panel's entries of A (row i, 16 columns) do not depend on step 1: their loads fly while the matrix cores work
            const int i = base + tid;
            const int ic = i < n ? i : n - 1;
            T pa[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) pa[c] = A[ic + (size_t)(c0 + c < n ? c0 + c : n - 1) * lda];
            // ---- 1. S = L[rows, :c0] L[c0:c0+16, :c0]^T for the 64 rows of this wave, on MFMA (skip rows above the panel).
            //      (Round 4 tried, one by one and with the stamps of scripts/solve_phases.py: row blocks dealt cyclically, the counted
            //      three-buffer ring of potrf_panel, 16-byte loads of row pairs, a branch-free loop body, per-lane offsets on a scalar
            //      base instead of a 64-bit multiply per address. The stage stayed at 400-450 us at n = 512 -- 2 x the matrix-core time
            //      of its busiest SIMD -- with every one of them, so the plain loop stays; what is left is spread over ring fills
            //      per panel and the two waves of a SIMD taking turns. Round 5, with the helpers leaving <= 3 panels to this loop: the
            //      first two panels' loads issued together ahead of the first matrix-core instruction made the stage SLOWER, 0.71 ->
            //      0.95 ms at n = 1024 -- 60 more live registers in a 512-thread workgroup.)
            T tv[16];
            if (useT) {
#pragma unroll
                for (int c = 0; c < 16; ++c) tv[c] = coop_ld(Tk + (size_t)(i >= c0 ? ic : n - 1) * kCoopBlockCols + c);
            }
            if (k > jstart && base + 64 * wave + 63 >= c0) {
                Acc acc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = Acc{0, 0, 0, 0};
                int rowa[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int r = base + 64 * wave + 16 * u + lr; rowa[u] = r < n ? r : n - 1; }
                const int rowb = c0 + lr < n ? c0 + lr : n - 1;
                for (int j = jstart; j < k; ++j) {
                    T fa[4][4], fb[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const size_t col = (size_t)(16 * j + 4 * s4 + lk) * ldf;
                        fb[s4] = F[rowb + col];
#pragma unroll
                        for (int u = 0; u < 4; ++u) fa[s4][u] = F[rowa[u] + col];
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[u] = Mma<T>::mma(fa[s4][u], fb[s4], acc[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        span[(64 * wave + 16 * u + Mma<T>::row(lane, r)) * 17 + lr] = acc[u][r];
            }
            __syncthreads();
            if (dbg && tid == 0) { const long long t = wall_clock64(); ph[0] += t - ph[3]; ph[3] = t; }
            // ---- 2. one row per thread: p = A[i, panel] - S[i, :]
            const bool active = i < n && i >= c0;
            T p[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const T v = pa[c];
                T sv = (k > jstart && i >= c0) ? span[tid * 17 + c] : T(0);
                if (useT && i >= c0) sv += tv[c];
                p[c] = (active && c0 + c < n) ? v - sv : ((i == c0 + c) ? T(1) : T(0));   // identity padding past n
            }
            const bool diag_chunk = base <= c0 && c0 < base + kBigThreads;
            if (diag_chunk) {
                // the 16 diagonal rows are one DPP row of wave wd: factor them with row broadcasts, publish L_kk and 1 / pivots
                const int wd = (c0 - base) >> 6, l0 = (c0 - base) & 63;
                if (wave == wd && lane >= l0 && lane < l0 + 16) {
                    const int r = lane - l0;
                    int bad = 0;
                    static_for<16>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        const T piv = dpp_row_bcast<c>(p[c]);
                        if (!(piv > 0)) { if (c0 + c < n && bad == 0) bad = c0 + c + 1; }
                        T rinv, d;
                        rsqrt_sqrt(piv > 0 ? piv : T(1), rinv, d);
                        if (r > c) p[c] *= rinv; else if (r == c) p[c] = d;
                        if (r == c) rd[c] = rinv;
                        static_for<16>([&](auto cc2) {
                            constexpr int c2 = decltype(cc2)::value;
                            if constexpr (c2 > c) {
                                const T lc = dpp_row_bcast<c2>(p[c]);
                                p[c2] -= p[c] * lc;          // rows r <= c: entries above the diagonal only, never read
                            }
                        });
                    });
#pragma unroll
                    for (int c = 0; c < 16; ++c) blk[r * 17 + c] = p[c];
                    if (bad != 0 && r == 0) sm.ired[kBigWaves] = bad;
                }
                __syncthreads();
                const int info = sm.ired[kBigWaves];
                if (info != 0) {                                // uniform
                    if (pending) coop_collect(*cc, kCoopPotrfT);
                    return info;
                }
            }
            if (dbg && tid == 0) { const long long t = wall_clock64(); ph[1] += t - ph[3]; ph[3] = t; }
            // ---- 3. the rows below the diagonal block solve against L_kk; store the panel
            if (active && i >= c0 + 16) {
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
                    if (c0 + c < n && i >= c0 + c) {
                        F[i + (size_t)(c0 + c) * ldf] = p[c];
                    }
            }
            __syncthreads();                                    // span reusable; the panel rows are visible
            if (dbg && tid == 0) { const long long t = wall_clock64(); ph[2] += t - ph[3]; }
        }
    }
    if (pending) coop_collect(*cc, kCoopPotrfT);             // (never: the last block publishes nothing)
    if (dbg && tid == 0) { dbg[16] = ph[0]; dbg[17] = ph[1]; dbg[18] = ph[2]; dbg[19] = ph_job; }
    return 0;
}

// ---------------------------------------------------------------- ?potrs: L L^T x = z, z in/out (global n-vector)
// The vector lives in LDS (sm.span) during the two sweeps. Block step kb: the 16 lanes of wave 0's first DPP row solve the
// diagonal block among themselves -- lane r holds row r (forward) / column r (backward) of L_kk and its reciprocal pivot; unknown
// after unknown: a multiply, a row broadcast, a multiply-add (0.4 us a block with its loads; before, ONE thread walked the 120
// dependent products through LDS: 3 us a block, 128 blocks a solve at n = 512) --, one barrier, then every remaining row subtracts
// its 16 products, the 16 factor entries loaded together (indices clamped: always 16 loads in flight). 330 -> 250 us a call at
// n = 512. (Tried on top and measured no better: the next step's loads issued ahead of the diagonal solve, and a transposed copy of
// the factor for a coalesced backward sweep -- stamps in profiles/r04/solve_big_phases_n512.txt.)
template <typename T>
__device__ __noinline__ void potrs_big(int n, const T* F, int ldf, T* z, BigLds<T>& sm)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int nblk = (n + 15) / 16;
    T* zs = sm.span;
    __syncthreads();
    for (int i = tid; i < n; i += kBigThreads) zs[i] = z[i];
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        const bool fwd = pass == 0;
        for (int kk = 0; kk < nblk; ++kk) {
            const int kb = fwd ? kk : nblk - 1 - kk;
            const int c0 = 16 * kb;
            if (tid < kWave) {                                   // wave 0; its lanes 16 .. 63 repeat the work of lanes 0 .. 15
                const int r = lane & 15;
                const bool live = c0 + r < n;
                const int gr = live ? c0 + r : n - 1;
                T lv[16];                                        // fwd: L[c0 + r][c0 + q], q < r; bwd: L[c0 + q][c0 + r], q > r; else 0
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int gq = c0 + q < n ? c0 + q : n - 1;
                    const T v = fwd ? F[gr + (size_t)gq * ldf] : F[gq + (size_t)gr * ldf];
                    lv[q] = ((fwd ? q < r : q > r) && live && c0 + q < n) ? v : T(0);
                }
                const T dg = F[gr + (size_t)gr * ldf];
                const T rinv = live ? T(1) / dg : T(1);
                T sv = live ? zs[c0 + r] : T(0), xr = 0;
                // lane q's candidate sv * rinv is x_q once every earlier unknown has been subtracted from its row
                if (fwd) {
                    static_for<16>([&](auto QQ) {
                        constexpr int q = decltype(QQ)::value;
                        const T xq = dpp_row_bcast<q>(sv * rinv);
                        xr = (r == q) ? xq : xr;
                        sv -= lv[q] * xq;
                    });
                } else {
                    static_for<16>([&](auto QQ) {
                        constexpr int q = 15 - decltype(QQ)::value;
                        const T xq = dpp_row_bcast<q>(sv * rinv);
                        xr = (r == q) ? xq : xr;
                        sv -= lv[q] * xq;
                    });
                }
                if (lane < 16) { if (live) zs[c0 + r] = xr; sm.xk[r] = live ? xr : T(0); }
            }
            __syncthreads();
            if (fwd) {
                for (int i = c0 + 16 + tid; i < n; i += kBigThreads) {
                    T lf[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) lf[c] = F[i + (size_t)(c0 + c) * ldf];          // c0 + c < n here: i > c0 + 15
                    T acc = 0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) acc += lf[c] * sm.xk[c];
                    zs[i] -= acc;
                }
            } else {
                for (int i = tid; i < c0; i += kBigThreads) {
                    T lf[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) lf[c] = F[(c0 + c < n ? c0 + c : n - 1) + (size_t)i * ldf];   // L(c0 + c, i)
                    T acc = 0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) if (c0 + c < n) acc += lf[c] * sm.xk[c];
                    zs[i] -= acc;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += kBigThreads) z[i] = zs[i];
    __syncthreads();
}

// ?potrs for n <= kBigThreads: ONE ROW PER THREAD, the scheme of potrs_rows (solve_kernel.h) on 512 threads. z stays in a register;
// block step kb: the 16 owners of rows 16 kb .. 16 kb + 15 (one DPP row of one wave) form x_kb = inv(L_kk) z_kb -- 16 independent
// products, no substitution chain -- and publish it; after ONE barrier every remaining row subtracts its 16 products, whose factor
// entries were loaded a step ahead (they do not depend on the solution). inv(L_kk): sm.dinv, filled once per factorisation
// (invert_diag_blocks). 245 -> ~80 us a call at n = 512 (potrs_big: a substitution chain, two barriers and two exposed L2 round trips
// per block step; it stays the routine for n > kBigThreads).
template <typename T, int RPT>
__device__ __noinline__ void potrs_big_rows(int n, const T* F_, int ldf, T* xv_, BigLds<T>& sm)
{
    // RPT rows per thread: row tid + kBigThreads r (RPT = 2: kBigThreads < n <= kBigRowsMaxN, 64 inverses in span + dinv)
    const gbl_cptr<T> F = as_global(F_);
    const gbl_ptr<T> xv = as_global_w(xv_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = (n + 15) / 16;
    lds_ptr<T> Dinv;
    if constexpr (RPT == 1) Dinv = as_lds(sm.dinv); else Dinv = as_lds(sm.span);
    const lds_ptr<T> xk = as_lds(sm.xk);
    __syncthreads();
    int ir[RPT], irc[RPT];
    T z[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) { ir[r] = tid + kBigThreads * r; irc[r] = ir[r] < n ? ir[r] : n - 1; z[r] = ir[r] < n ? xv[ir[r]] : T(0); }
    T lnext[RPT][16];
    // (a wave none of whose rows takes part in a step skips that step's loads -- half of them on average: eight waves' 16 loads a
    //  step are 0.85 us of the CU's vector-memory address rate)
    auto load_row = [&](int kb) {                              // L[i][16 kb .. 16 kb + 15], used by rows i >= 16 kb + 16
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (kBigThreads * r + 64 * wave + 63 >= 16 * kb + 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) lnext[r][c] = F[irc[r] + (size_t)(16 * kb + c < n ? 16 * kb + c : n - 1) * ldf];
            }
    };
    auto load_col = [&](int kb) {                              // L[16 kb .. 16 kb + 15][i], used by rows i < 16 kb
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (kBigThreads * r + 64 * wave < 16 * kb) {
#pragma unroll
                for (int c = 0; c < 16; ++c) lnext[r][c] = F[(16 * kb + c < n ? 16 * kb + c : n - 1) + (size_t)irc[r] * ldf];
            }
    };
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
        for (int c = 0; c < 16; ++c) lnext[r][c] = 0;
    // the 16 owners of rows 16 kb .. 16 kb + 15 (one DPP row of one wave, row slot rs) form x_kb = op(inv(L_kk)) z_kb
    auto diag = [&](int kb, bool transposed) {
        const int c0 = 16 * kb, rs = c0 / kBigThreads, w0 = (c0 % kBigThreads) >> 6;
        if (wave == w0) {
            const int l0 = c0 & 63, r = lane & 15;
            T zz = z[0];
#pragma unroll
            for (int q = 1; q < RPT; ++q) zz = (rs == q) ? z[q] : zz;
            T dv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) dv[c] = transposed ? Dinv[kb * 272 + c + 17 * r] : Dinv[kb * 272 + r + 17 * c];
            T xn = 0;
            static_for<16>([&](auto cc) { constexpr int c = decltype(cc)::value; xn += dv[c] * dpp_row_bcast<c>(zz); });
            if (lane >= l0 && lane < l0 + 16) {
#pragma unroll
                for (int q = 0; q < RPT; ++q) z[q] = (rs == q) ? xn : z[q];
                xk[(kb & 1) * 16 + r] = xn;
            }
        }
    };
    // ---- forward: L z = b
    load_row(0);
    for (int kb = 0; kb < nb; ++kb) {
        const int c0 = 16 * kb;
        T lrow[RPT][16];
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int c = 0; c < 16; ++c) lrow[r][c] = lnext[r][c];
        if (kb + 1 < nb) load_row(kb + 1); else load_col(nb - 1);
        diag(kb, false);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (ir[r] >= c0 + 16 && ir[r] < n) {
                T acc = 0;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc += lrow[r][c] * ((c0 + c < n) ? xk[(kb & 1) * 16 + c] : T(0));
                z[r] -= acc;
            }
    }
    // ---- backward: L^T x = z
    for (int kb = nb - 1; kb >= 0; --kb) {
        const int c0 = 16 * kb;
        T lcol[RPT][16];
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int c = 0; c < 16; ++c) lcol[r][c] = lnext[r][c];
        if (kb > 0) load_col(kb - 1);
        diag(kb, true);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (ir[r] < c0) {
                T acc = 0;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc += lcol[r][c] * ((c0 + c < n) ? xk[(kb & 1) * 16 + c] : T(0));
                z[r] -= acc;
            }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        if (ir[r] < n) xv[ir[r]] = z[r];
    __syncthreads();
}

// inverses of the 16 x 16 diagonal blocks of the factor, any number of blocks (one thread per (block, column), strided)
template <typename T>
__device__ __noinline__ void invert_diag_blocks_big(int n, const T* F_, int ldf, T* Dinv_)
{
    const gbl_cptr<T> F = as_global(F_);
    const lds_ptr<T> Dinv = as_lds(Dinv_);
    const int nb = (n + 15) / 16;
    for (int e = threadIdx.x; e < 16 * nb; e += kBigThreads) {
        const int k = e >> 4, c = e & 15;
        const int base = 16 * k;
        T x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            T sacc = (r == c) ? T(1) : T(0);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < r) {
                    const int ir = base + r < n ? base + r : n - 1, iq = base + q < n ? base + q : n - 1;
                    sacc -= F[ir + (size_t)iq * ldf] * x[q];
                }
            const int ir = base + r < n ? base + r : n - 1;
            const T dr = F[ir + (size_t)ir * ldf];
            x[r] = (base + r < n) ? ((r >= c) ? sacc / dr : T(0)) : ((r == c) ? T(1) : T(0));   // identity past n
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Dinv[k * 272 + r + 17 * c] = x[r];
    }
    __syncthreads();
}

// r = b - A x, w = |b| + |A| |x|: one ROW per thread (rows strided by the workgroup), A symmetric: entry (i, k) is read as
// A[i + k lda], consecutive rows in consecutive lanes; 16 columns' loads are issued together (indices clamped: always 16, a count
// the compiler can keep in flight) and x comes from LDS. (Before: one wave per row with a wave reduction per row -- 64 dependent
// round trips a wave, 115 us per residual at n = 512.)
template <typename T>
__device__ __noinline__ void residual_big(int n, const T* A_, int lda, const T* b, const T* x, T* r, T* w, T* xs_ /* LDS, >= n */,
                                          CoopCtx* cc = nullptr, T* part = nullptr)
{
    if (cc && cc->W > 1 && part && !cc->failed) {
        // helpers: the columns in W contiguous ranges, one per workgroup (solve_coop.h, kCoopSymv); a row's W partial sums are
        // added in peer order
        const unsigned long long d[6] = {(unsigned long long)(uintptr_t)A_, (unsigned long long)(uintptr_t)x, (unsigned long long)(uintptr_t)part,
                                         (unsigned long long)(unsigned)n | ((unsigned long long)(unsigned)lda << 32), 1ull, 0ull};
        coop_publish(*cc, kCoopSymv | kCoopFence, d);
        coop_symv<T>(d, 0, cc->W, xs_);
        coop_collect(*cc, kCoopSymv | kCoopFence);
        if (!cc->failed) {                                      // (a helper that timed out: the sweep below, on this workgroup alone)
            for (int i = threadIdx.x; i < n; i += kBigThreads) {
                T ra = 0, wa = 0;
                for (int p = 0; p < cc->W; ++p) { ra += part[(size_t)(2 * p) * n + i]; wa += part[(size_t)(2 * p + 1) * n + i]; }
                r[i] = b[i] - ra;
                w[i] = dabs(b[i]) + wa;
            }
            __syncthreads();
            return;
        }
    }
    const gbl_cptr<T> A = as_global(A_);
    const lds_ptr<T> xs = as_lds(xs_);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kBigThreads) xs[i] = x[i];
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += kBigThreads) {
        const int i = i0 + threadIdx.x;
        const int ic = i < n ? i : n - 1;
        T ra = 0, wa = 0;
        // two batches of 16 columns: the next one is in flight while this one is multiplied (loads unconditional, indices clamped;
        // the column guard sits at the use -- the if-around-a-load rule of DESIGN section 3.2)
        T av[2][16];
        auto issue = [&](int k0, auto B) {
            constexpr int buf = decltype(B)::value;
#pragma unroll
            for (int u = 0; u < 16; ++u) av[buf][u] = A[ic + (size_t)(k0 + u < n ? k0 + u : n - 1) * lda];
        };
        auto use = [&](int k0, auto B) {
            constexpr int buf = decltype(B)::value;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + u < n) { const T xv = xs[k0 + u]; ra += av[buf][u] * xv; wa += dabs(av[buf][u]) * dabs(xv); }
        };
        issue(0, IntC<0>{});
        for (int k0 = 0; k0 < n; k0 += 32) {
            issue(k0 + 16, IntC<1>{});
            use(k0, IntC<0>{});
            issue(k0 + 32, IntC<0>{});
            use(k0 + 16, IntC<1>{});
        }
        if (i < n) { r[i] = b[i] - ra; w[i] = dabs(b[i]) + wa; }
    }
    __syncthreads();
}

// which ?potrs: one / two rows per thread with the diagonal blocks' inverses in LDS, or (0) the block-step routine.
// (Two rows per thread: double only -- the float build of that instance trips an instruction-selection error of hipcc 7.2,
//  "V_CMP_NE_U32 0, $src_shared_base"; so does a lambda around these calls. float above n = 512 keeps the block-step routine.)
template <typename T> __device__ __forceinline__ int potrs_rows_per_thread(int n)
{
    return n <= kBigThreads ? 1 : ((sizeof(T) == 8 && n <= kBigRowsMaxN) ? 2 : 0);
}
template <typename T>
__device__ __forceinline__ void potrs_any(int rows, int n, const T* F, int ldf, T* v, BigLds<T>& sm)
{
    if (rows == 1) potrs_big_rows<T, 1>(n, F, ldf, v, sm);
    else if (rows == 2) { if constexpr (sizeof(T) == 8) potrs_big_rows<T, 2>(n, F, ldf, v, sm); }
    else potrs_big<T>(n, F, ldf, v, sm);
}

// ---------------------------------------------------------------- ?posvx('E','L'), nrhs = 1 (same semantics as posvx_device)
template <typename T>
__device__ __noinline__ int posvx_big(int n, T* A, int lda, T* F, int ldf, T* s, T* b, T* x, T* r, T* w, BigLds<T>& sm, long long* dbg = nullptr,
                                      CoopCtx* cc = nullptr, T* cS = nullptr)
{
    MIRLSQ_STAMP(dbg, 2);
    const int tid = threadIdx.x;
    const T eps = Lim<T>::eps / 2;
    const T safmin = Lim<T>::min_normal;
    // ?poequ
    T mn = Lim<T>::inf(), mx = -Lim<T>::inf();
    for (int i = tid; i < n; i += kBigThreads) { const T d = A[i + (size_t)i * lda]; mn = d < mn ? d : mn; mx = d > mx ? d : mx; }
    const T smin = big_min(mn, sm.red);
    const T amax = big_max(mx, sm.red);
    bool rcequ = false;
    if (smin > 0) {
        const T scond = dsqrt(smin) / dsqrt(amax);
        for (int i = tid; i < n; i += kBigThreads) s[i] = T(1) / dsqrt(A[i + (size_t)i * lda]);
        const T small = safmin / Lim<T>::eps, large = T(1) / small;
        rcequ = !(scond >= T(0.1) && amax >= small && amax <= large);
    }
    __syncthreads();
    if (rcequ) {                                                // ?laqsy
        for (int j = tid >> 6; j < n; j += kBigWaves) {
            const T cj = s[j];
            for (int i = tid & 63; i < n; i += kWave) A[i + (size_t)j * lda] = cj * s[i] * A[i + (size_t)j * lda];
        }
        for (int i = tid; i < n; i += kBigThreads) b[i] = s[i] * b[i];
    }
    __syncthreads();
    MIRLSQ_STAMP(dbg, 3);
    const int info = potrf_big<T>(n, A, lda, F, ldf, sm, dbg, cc, cS);
    if (info != 0) return info;
    MIRLSQ_STAMP(dbg, 4);
    // uniform: one or two rows per thread and the inverses of the diagonal blocks in LDS (n <= 1024), else the block-step routine
    // (two rows per thread: double only -- the float build of that instance trips an instruction-selection error of hipcc 7.2,
    //  "V_CMP_NE_U32 0, $src_shared_base"; float above n = 512 keeps the block-step routine)
    const int rows = potrs_rows_per_thread<T>(n);
    // (no select between two LDS arrays here or below: hipcc 7.2 fails to select the address-space cast of one in the float build)
    if (rows == 1) invert_diag_blocks_big<T>(n, F, ldf, sm.dinv);
    else if (rows == 2) invert_diag_blocks_big<T>(n, F, ldf, sm.span);
    for (int i = tid; i < n; i += kBigThreads) x[i] = b[i];
    potrs_any<T>(rows, n, F, ldf, x, sm);
    MIRLSQ_STAMP(dbg, 5);
    // ?porfs, ITMAX = 5
    const T safe1 = T(n + 1) * safmin, safe2 = safe1 / eps;
    T lstres = 3;
    for (int count = 1;; ++count) {
        if (n <= kBigRowsMaxN) residual_big<T>(n, A, lda, b, x, r, w, sm.xs, cc, cS ? cS + coop_part_offset(n) : nullptr);
        else residual_big<T>(n, A, lda, b, x, r, w, sm.span, cc, cS ? cS + coop_part_offset(n) : nullptr);
        if (count == 1) MIRLSQ_STAMP(dbg, 11);
        T qv = 0;
        for (int i = tid; i < n; i += kBigThreads) {
            const T q = (w[i] > safe2) ? dabs(r[i]) / w[i] : (dabs(r[i]) + safe1) / (w[i] + safe1);
            qv = q > qv ? q : qv;
        }
        const T berr = big_max(qv, sm.red);
        if (count == 1) MIRLSQ_STAMP(dbg, 12);
        if (berr > eps && 2 * berr <= lstres && count <= 5) {
            potrs_any<T>(rows, n, F, ldf, r, sm);
            for (int i = tid; i < n; i += kBigThreads) x[i] += r[i];
            lstres = berr;
            __syncthreads();
            continue;
        }
        break;
    }
    if (rcequ) for (int i = tid; i < n; i += kBigThreads) x[i] = s[i] * x[i];
    __syncthreads();
    MIRLSQ_STAMP(dbg, 6);
    return 0;
}

// ---------------------------------------------------------------- solveBoxQP, boxcqp.d:122-379 (loops over n)
// A (n x n) must hold the matrix of the unconstrained system on entry (it is overwritten); Pm stays intact.
template <typename T>
__device__ __noinline__ int box_qp_big(int n, const T* Pm, const T* q, const T* l, const T* u, T* x, bool unconstrainedSolution,
                                       T relTol, T absTol, uint32_t maxIterations, SolveScratch<T>& sc, BigLds<T>& sm, int* iters,
                                       CoopCtx* cc = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T* s = sc.vec;
    T* b = sc.vec + 1 * (size_t)n;
    T* r = sc.vec + 2 * (size_t)n;
    T* w = sc.vec + 3 * (size_t)n;
    T* la = sc.vec + 4 * (size_t)n;
    T* mu = sc.vec + 5 * (size_t)n;
    T* sX = sc.vec + 6 * (size_t)n;
    int32_t* SI = sc.ivec;
    int32_t* flags = sc.ivec + n;
    const int ldf = n | 1;
    *iters = 0;
    if (n == 0) return 0;

    if (!unconstrainedSolution) {                                   // QP:168-214
        for (int i = tid; i < n; i += kBigThreads) b[i] = -q[i];
        __syncthreads();
        const int info = posvx_big<T>(n, sc.A, n, sc.Fg, ldf, s, b, x, r, w, sm, sc.dbg, cc, sc.cS);
        if (info != 0) return 1;
    }
    {                                                               // QP:216-219
        int bad = 0;
        for (int i = tid; i < n; i += kBigThreads) bad |= !(l[i] <= x[i] && x[i] <= u[i]);
        if (!big_or(bad, sm.ired)) return 0;
    }
    if (!maxIterations) maxIterations = (uint32_t)n * 10 + 100;     // QP:224-226
    for (int i = tid; i < n; i += kBigThreads) { la[i] = 0; mu[i] = 0; }
    __syncthreads();

    for (uint32_t step = 0; step < maxIterations; ++step) {         // QP:234
        *iters = (int)step + 1;
        // classification QP:239-263; the free-set list SI keeps the reference's ascending order (chunked ballot prefix sum)
        int sN = 0;
        for (int base = 0; base < n; base += kBigThreads) {
            const int i = base + tid;
            int fl = 2;
            if (i < n) {
                const T xi = x[i], li = l[i], ui = u[i];
                const T xl = xi - li, ux = ui - xi;
                if (xl < 0 || (xl < relTol + absTol * dabs(li) && la[i] >= 0)) { fl = -1; x[i] = li; mu[i] = 0; }
                else if (ux < 0 || (ux < relTol + absTol * dabs(ui) && mu[i] >= 0)) { fl = 1; x[i] = ui; la[i] = 0; }
                else { fl = 0; mu[i] = 0; la[i] = 0; }
                flags[i] = fl;
            }
            const unsigned long long bal = __ballot(fl == 0);
            __syncthreads();
            if (lane == 0) sm.ired[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int wv = 0; wv < kBigWaves; ++wv) { if (wv < wave) before += sm.ired[wv]; total += sm.ired[wv]; }
            if (fl == 0) SI[sN + before + __popcll(bal & ((1ull << lane) - 1))] = i;
            sN += total;
        }
        __syncthreads();
        if (sN == n) break;                                         // QP:265-266 (quirk Q8)

        // reduced system QP:282-305, Kahan-Babuska-Neumaier sums over the bound variables, j ascending
        for (int ii = tid; ii < sN; ii += kBigThreads) {
            const int i = SI[ii];
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
                    sc.A[(size_t)jj * sN + ii] = pij;
                    ++jj;
                }
            }
            b[ii] = -(ks + kc);
        }
        __syncthreads();
        if (sN) {                                                   // QP:307-325
            const int info = posvx_big<T>(sN, sc.A, sN, sc.Fg, sN | 1, s, b, sX, r, w, sm, nullptr, cc, sc.cS);
            if (info != 0) return 1;
        }
        for (int ii = tid; ii < sN; ii += kBigThreads) x[SI[ii]] = sX[ii];   // QP:327-329
        __syncthreads();
        // multipliers QP:333-337
        for (int i = tid; i < n; i += kBigThreads) {
            if (!flags[i]) continue;
            T v1 = 0, v2 = 0;
            for (int j = 0; j < i; ++j) v1 += Pm[(size_t)j * n + i] * x[j];
            for (int j = i; j < n; ++j) v2 += Pm[(size_t)j * n + i] * x[j];
            const T val = v1 + v2 + q[i];
            if (flags[i] < 0) la[i] = val; else mu[i] = -val;
        }
        __syncthreads();
        int again = 0;                                              // QP:339-347
        for (int i = tid; i < n; i += kBigThreads) {
            const int fi = flags[i];
            if (fi < 0) again |= !(la[i] >= 0);
            else if (fi > 0) again |= !(mu[i] >= 0);
            else again |= !(x[i] >= l[i] && x[i] <= u[i]);
        }
        if (big_or(again, sm.ired)) continue;
        for (int i = tid; i < n; i += kBigThreads) x[i] = dfmax(dfmin(x[i], u[i]), l[i]);   // QP:349
        __syncthreads();
        return 0;
    }
    return 2;                                                       // QP:378
}

// ---------------------------------------------------------------- one LM pass, n x n part, any n
template <typename T>
__device__ __forceinline__ void lm_solve_big_main(const LmSolveArgs<T>& a, int kc, BigLds<T>& sm, CoopCtx* cc)
{
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SolveScratch<T> sc = a.sc[kc];
    T* dx_out = a.dx + (size_t)kc * n;
    T* trial_out = a.trial + (size_t)kc * n;
    T* qpl = sc.vec + 7 * (size_t)n;
    T* qpu = sc.vec + 8 * (size_t)n;
    T* tv = sc.vec + 9 * (size_t)n;                  // JJ dx for the predicted reduction
    T* xq = sc.vec + 10 * (size_t)n;

    MIRLSQ_STAMP(sc.dbg, 0);
    if (sc.dbg && threadIdx.x == 0) sc.dbg[9] = clock64();
    T jy_inf = 0;
    if (a.check_grad) {
        T mx = 0;
        for (int i = tid; i < n; i += kBigThreads) { const T av = dabs(a.Jy[i]); if (av > mx) mx = av; }
        jy_inf = big_max(mx, sm.red);
        __syncthreads();
    }
    if (a.check_grad && tid == 0 && kc == 0) a.st->jy_inf = jy_inf;
    if (a.check_grad && !(jy_inf > a.set.gradTolerance)) {          // LS:1053
        if (tid == 0) { ChainRec<T> r{}; r.flags = kFlagGradSmall; a.rec[kc] = r; }
        return;
    }
    // lambda_0, LS:1067-1072: the FIRST diagonal entry of maximum modulus, as i?amax picks it
    T lambda = (kc == 0 && (a.lambda_from_state || a.lambda_from_device)) ? a.st->lambda : a.lam[kc];
    if (kc == 0 && a.lambda_from_state && !(lambda >= a.set.minLambda)) {
        T best = -1;
        int where = 0x7fffffff;
        for (int i = tid; i < n; i += kBigThreads) {
            const T dg = dabs(a.JJ[(size_t)i * n + i]);
            if (dg > best) { best = dg; where = i; }             // ascending i per thread: first maximum
        }
        const T mx = big_max(best, sm.red);
        int cand = (best == mx) ? where : 0x7fffffff;
#pragma unroll
        for (int k = 1; k < kWave; k <<= 1) { const int o = __shfl_xor(cand, k, kWave); cand = o < cand ? o : cand; }
        __syncthreads();
        if (lane == 0) sm.ired[wave] = cand;
        __syncthreads();
        int first = sm.ired[0];
        for (int wv = 1; wv < kBigWaves; ++wv) first = sm.ired[wv] < first ? sm.ired[wv] : first;
        lambda = T(0.001) * a.JJ[(size_t)first * n + first];
        if (!(lambda >= a.set.minLambda)) lambda = 1;
        __syncthreads();
    }
    // step bounds LS:1074-1077; Pm = A = JJ + lambda I, LS:1078-1079
    for (int i = tid; i < n; i += kBigThreads) { qpl[i] = a.lower[i] - a.x[i]; qpu[i] = a.upper[i] - a.x[i]; }
    if (cc && !cc->failed) {
        // helpers: the two copies of J^T J over all workgroups of the entry (0.36 ms on one at n = 1024)
        const unsigned long long d[6] = {(unsigned long long)(uintptr_t)a.JJ, (unsigned long long)(uintptr_t)sc.Pm, (unsigned long long)(uintptr_t)sc.A,
                                         (unsigned long long)n * (unsigned long long)n, 0ull, 0ull};
        coop_publish(*cc, kCoopCopy2 | kCoopFence, d);
        coop_copy2<T>(d, 0, cc->W);
        coop_collect(*cc, kCoopCopy2 | kCoopFence);
    }
    {
        // 16 loads in flight a thread; the damping goes onto the diagonal in a pass of its own (an `idx % (n + 1)` per element was
        // a 64-bit division per element: 205 us of this kernel at n = 512)
        const size_t nn = (cc && !cc->failed) ? 0 : (size_t)n * n;
        for (size_t base = tid; base < nn; base += (size_t)16 * kBigThreads) {
            T v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const size_t idx = base + (size_t)u * kBigThreads; v[u] = a.JJ[idx < nn ? idx : nn - 1]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const size_t idx = base + (size_t)u * kBigThreads;
                if (idx < nn) { sc.Pm[idx] = v[u]; sc.A[idx] = v[u]; }
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += kBigThreads) {
            const T t = a.JJ[(size_t)i * n + i] + lambda;
            sc.Pm[(size_t)i * n + i] = t;
            sc.A[(size_t)i * n + i] = t;
        }
    }
    __syncthreads();
    MIRLSQ_STAMP(sc.dbg, 1);
    int qp_iters = 0;
    const int qp = box_qp_big<T>(n, sc.Pm, a.Jy, qpl, qpu, xq, false, a.set.qpRelTolerance, a.set.qpAbsTolerance,
                                 a.set.qpMaxIterations, sc, sm, &qp_iters, cc);   // LS:1080
    MIRLSQ_STAMP(sc.dbg, 7);

    int flags = 0;
    T ndd = 0, pred = 0, xn = 0;
    if (qp == 0) {
        int fl = 0, moved = 0;
        T sdd = 0, amx = 0;
        for (int i = tid; i < n; i += kBigThreads) {
            T d = xq[i];
            if (!(d <= d)) fl |= kFlagDxNaN;                         // LS:1087
            const T xi = a.x[i];
            d = d + xi;                                              // LS:1096
            d = d - xi;                                              // LS:1097
            dx_out[i] = d;
            const T tr = dfmax(dfmin(d + xi, a.upper[i]), a.lower[i]);   // LS:1108-1110
            trial_out[i] = tr;
            if (!(tr <= tr)) fl |= kFlagXNaN;
            moved |= !(tr == xi);
            sdd += d * d;
            const T at = dabs(tr);
            amx = at > amx ? at : amx;
        }
        if (big_or(fl & kFlagDxNaN, sm.ired)) flags |= kFlagDxNaN;
        if (big_or(fl & kFlagXNaN, sm.ired)) flags |= kFlagXNaN;
        if (!big_or(moved, sm.ired)) flags |= kFlagNullStep;
        ndd = big_sum(sdd, sm.red);                                  // LS:1099
        // predicted reduction with the UNDAMPED JJ, LS:1141-1142: t = JJ dx + 2 Jy ; pred = -(t . dx)
        // (JJ dx)_i with one row per thread: JJ is symmetric, entry (i, k) is read as JJ[k n + i] -- consecutive rows in consecutive
        // lanes --, dx from LDS, 16 loads in flight (before: a wave per row and a wave reduction per row, 160 us at n = 512)
        __syncthreads();
        if (cc && !cc->failed) {
            T* part = sc.cS + coop_part_offset(n);
            const unsigned long long d[6] = {(unsigned long long)(uintptr_t)a.JJ, (unsigned long long)(uintptr_t)dx_out, (unsigned long long)(uintptr_t)part,
                                             (unsigned long long)(unsigned)n | ((unsigned long long)(unsigned)n << 32), 0ull, 0ull};
            coop_publish(*cc, kCoopSymv | kCoopFence, d);
            coop_symv<T>(d, 0, cc->W, sm.span);
            coop_collect(*cc, kCoopSymv | kCoopFence);
            if (!cc->failed)
                for (int i = tid; i < n; i += kBigThreads) {
                    T acc = 0;
                    for (int p = 0; p < cc->W; ++p) acc += part[(size_t)(2 * p) * n + i];
                    tv[i] = acc;
                }
        }
        for (int i = tid; i < n; i += kBigThreads) sm.span[i] = dx_out[i];
        __syncthreads();
        for (int i0 = 0; !(cc && !cc->failed) && i0 < n; i0 += kBigThreads) {     // (no helpers, or one timed out: here)
            const int i = i0 + tid, ic = i < n ? i : n - 1;
            T acc = 0;
            for (int k0 = 0; k0 < n; k0 += 16) {
                T av[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) av[u] = a.JJ[(size_t)(k0 + u < n ? k0 + u : n - 1) * n + ic];
#pragma unroll
                for (int u = 0; u < 16; ++u) if (k0 + u < n) acc += av[u] * sm.span[k0 + u];
            }
            if (i < n) tv[i] = acc;
        }
        __syncthreads();
        T sp = 0;
        for (int i = tid; i < n; i += kBigThreads) sp += (tv[i] + 2 * a.Jy[i]) * dx_out[i];
        pred = -big_sum(sp, sm.red);
        const T amax = big_max(amx, sm.red);                         // LS:1164, scaled like ?nrm2
        T sc2 = 0;
        if (amax > 0)
            for (int i = tid; i < n; i += kBigThreads) { const T v = trial_out[i] / amax; sc2 += v * v; }
        xn = amax > 0 ? amax * dsqrt(big_sum(sc2, sm.red)) : T(0);
        if (!(dsqrt(ndd) < a.set.maxStep)) flags |= kFlagStepTooLong; // LS:1101
    }
    MIRLSQ_STAMP(sc.dbg, 8);
    if (sc.dbg && threadIdx.x == 0) sc.dbg[10] = clock64();
    if (tid == 0) {
        ChainRec<T> r{};
        r.lambda = lambda; r.new_dx_dot = ndd; r.predicted = pred; r.trial_xnorm = xn;
        // a helper that did not answer within kCoopSpinSeconds: the entry is marked for the RESCUE launch that follows (it solves
        // the entry again from its inputs on one workgroup, once every workgroup of this launch -- late helpers included -- has
        // ended): a stall is a scheduling fact, the caller gets the one-workgroup answer and a count in its statistics
        r.qp_status = (cc && cc->failed) ? kQpCoopTimeout : qp; r.qp_iterations = qp_iters; r.flags = flags;
        a.rec[kc] = r;
    }
}

// One workgroup per ladder entry, or -- a.coop_w > 1 -- that workgroup plus coop_w - 1 helpers (solve_coop.h): block
// kc * coop_w is entry kc's main workgroup, the coop_w - 1 behind it serve its jobs until it says kCoopExit.
template <typename T>
__global__ __launch_bounds__(kBigThreads) void k_lm_solve_big(LmSolveArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];     // sizeof(BigLds<T>) > 64 KB: dynamic
    BigLds<T>& sm = *reinterpret_cast<BigLds<T>*>(big_smem);
    __shared__ int s_cflag;
    __shared__ unsigned long long s_cdesc[8];
    const int W = a.coop_w > 1 ? a.coop_w : 1;
    const int kc = (int)blockIdx.x / W, peer = (int)blockIdx.x % W;
    if (a.coop_rescue) {                                         // (W == 1)
        if (threadIdx.x == 0) s_cflag = a.rec[kc].qp_status == kQpCoopTimeout;
        __syncthreads();
        if (!s_cflag) return;
        __syncthreads();
        lm_solve_big_main<T>(a, kc, sm, nullptr);
        if (threadIdx.x == 0) a.rec[kc].flags |= kFlagCoopRescued;
        return;
    }
    if (peer > 0) {
        coop_helper_loop<T>(a.sc[kc].coop, W, peer, a.coop_epoch, sm.span);
        return;
    }
    CoopCtx cc{a.sc[kc].coop, W, 0, a.coop_epoch, 0u, 0, &s_cflag, s_cdesc};
    lm_solve_big_main<T>(a, kc, sm, W > 1 ? &cc : nullptr);
    if (W > 1) {
        const unsigned long long d[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
        coop_publish(cc, kCoopExit, d);
    }
}

// standalone BOXCQP for any n (mir_solve_box_qp_gpu_* above n = 256)
template <typename T>
__global__ __launch_bounds__(kBigThreads) void k_box_qp_big(BoxQpArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    BigLds<T>& sm = *reinterpret_cast<BigLds<T>*>(big_smem);
    const int n = a.n;
    const size_t nn = (size_t)n * n;
    // symmetrise the lower triangle (QP:109: only the lower triangle of P is meaningful); A = Pm
    for (size_t idx = threadIdx.x; idx < nn; idx += kBigThreads) {
        const size_t i = idx / n, j = idx % n;
        const T v = i >= j ? a.P[i * n + j] : a.P[j * n + i];
        a.sc.Pm[idx] = v;
        a.sc.A[idx] = v;
    }
    __syncthreads();
    int it = 0;
    const int st = box_qp_big<T>(n, a.sc.Pm, a.q, a.l, a.u, a.x, a.unconstrained != 0, a.relTol, a.absTol, a.maxIterations,
                                 a.sc, sm, &it);
    if (threadIdx.x == 0) { a.out[0] = st; a.out[1] = it; }
}

}  // namespace mirlsq
