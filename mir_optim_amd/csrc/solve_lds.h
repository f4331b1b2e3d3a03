// solve_lds.h -- ?posvx('E','L'), one right-hand side, entirely in LDS for n <= 128 (gfx950): the latency path of k_lm_solve.
//
// Replaces mir-lapack's posvx as the reference calls it (boxcqp.d:194, 310; Netlib ?posvx = ?poequ + ?laqsy + ?potrf +
// ?potrs + ?porfs; ?pocon and the forward-error bound are not computed: they only feed rcond / ferr, which the reference
// ignores, boxcqp.d:212, 323). Round 1's path (potrf_tiled2 / potrs_blocked) took ~150 us at
// n = 128 -- 13 % of a cfg-3 solve on one CU and most of a strong-scaled one: a column-by-column Cholesky (1 100 cycles
// per column), one-wave triangular solves built on v_readlane broadcasts, and refinement mat-vecs that fetched the matrix
// from L2 one dependent round trip at a time. This one keeps BOTH the factor and the matrix in LDS (74 us at n = 128):
//
//   * storage: the lower BLOCK triangle of 16 x 16 blocks, block (I, J), J <= I, at ((I (I + 1) / 2 + J) * 272), element
//     (r, c) at r + 17 c. The leading dimension 17 makes row walks, column walks and the MFMA operand pattern
//     (r = lane & 15, c = 4 s + (lane >> 4)) all bank-conflict-free. n = 128: 36 blocks = 78 KB per matrix, 157 KB for
//     L and A together plus 3 KB of vectors: fits the 160 KB of a CU's LDS;
//   * ?potrf: right-looking by 16-column panels with look-ahead, two barriers per panel: (1) one thread per row below the
//     diagonal block solves its 16 panel entries against L_kk, (2) wave 0 brings block (k + 1, k + 1) up to date and factors
//     it -- in the MFMA accumulator layout: four-column panels on the VALU (DPP row broadcasts, v_rsq_f64 + Goldschmidt),
//     rank-4 updates as one MFMA each, operands moved between lane groups by permlane swaps -- WHILE waves 1..3 apply
//     panel k to the rest of the trailing matrix (v_mfma_f64_16x16x4, operands and accumulators straight from / to the LDS
//     blocks, two independent chains per wave);
//   * the inverses of the diagonal blocks are stored TRANSPOSED IN THE UNUSED UPPER TRIANGLES of the diagonal blocks of L
//     (their diagonals are the reciprocal pivots, kept in a vector): no extra storage;
//   * ?potrs: ONE wave, no workgroup barrier inside: lane (row r of the block, quarter h of its columns), the vector solved
//     in place in LDS block by block, fully unrolled and software-pipelined (the products with finished blocks are summed
//     while the current block's 16-value exchange is in flight); wave_lds_fence() keeps the compiler from reusing values
//     other lanes have overwritten;
//   * ?porfs: mat-vec with two threads per row from the LDS copy of A (equilibrated when ?laqsy says so), same berr
//     test and ITMAX = 5 as Netlib.
// Rows and columns past n are an identity extension, so no routine needs row masks.
#pragma once

#include "common.h"
#include "solve_types.h"

namespace mirlsq {


template <int NB> struct LdsSolveCfg {
    static constexpr int NBT = NB * (NB + 1) / 2;
    static constexpr int NV = 16 * NB;
    static constexpr int L_OFF = 0;
    static constexpr int A_OFF = NBT * kLdsBlk;
    static constexpr int RD_OFF = 2 * NBT * kLdsBlk;     // reciprocal pivots
    static constexpr int XV_OFF = RD_OFF + NV;           // published solution block / x for the mat-vec
    static constexpr int ZV_OFF = XV_OFF + NV;           // exchange vector (scales, residual)
    static constexpr int ZERO_OFF = ZV_OFF + NV;         // one element that holds 0 (masked coefficient reads)
    static constexpr int ELEMS = ZERO_OFF + 2;
};

__device__ __forceinline__ int blk_off(int I, int J, int r, int c) { return (I * (I + 1) / 2 + J) * kLdsBlk + r + 17 * c; }

// ---- load: L = A = src (+ shift on the diagonal), identity past n. src: n x n full symmetric, leading dimension ld.
//      Two halves: lds_load_issue puts the global loads in flight (a thread's NBT = NB (NB + 1) / 2 values, one per block, and
//      the diagonal entry of row tid), lds_load_commit adds the shift and writes the LDS blocks. k_lm_solve issues the loads
//      of J^T J at kernel entry, BEFORE its prologue (|J^T y|_inf, lambda_0: dependent global loads and workgroup reductions of
//      their own), so that the memory latency (J^T J was written by another kernel, on other XCDs: it comes from the memory
//      side, not from this CU's L2) runs behind the prologue instead of being waited for inside ?posvx.
template <typename T, int NB> struct LdsPreload {
    T v[LdsSolveCfg<NB>::NBT];
    T diag;                                                  // src[tid][tid] (tid < n), unshifted
};
template <typename T, int NB>
__device__ __forceinline__ void lds_load_issue(int n, const T* __restrict__ src, int ld, LdsPreload<T, NB>& p)
{
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;      // lanes along a row of src: coalesced reads
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            const int gi = 16 * I + r, gj = 16 * J + c;
            const bool in = gi < n && gj < n;
            p.v[I * (I + 1) / 2 + J] = src[(size_t)(in ? gi : 0) * ld + (in ? gj : 0)];
        }
    const int t = (int)threadIdx.x < n ? (int)threadIdx.x : 0;
    p.diag = src[(size_t)t * ld + t];
}
template <typename T, int NB>
__device__ __forceinline__ void lds_load_commit(int n, const LdsPreload<T, NB>& p, T shift, T* smem)
{
    using C = LdsSolveCfg<NB>;
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            const int gi = 16 * I + r, gj = 16 * J + c;
            const bool in = gi < n && gj < n;
            const T t = p.v[I * (I + 1) / 2 + J];
            const T v = in ? (gi == gj ? t + shift : t) : (gi == gj ? T(1) : T(0));
            const int o = blk_off(I, J, r, c);
            smem[C::L_OFF + o] = v;
            smem[C::A_OFF + o] = v;
        }
    if (threadIdx.x == 0) smem[C::ZERO_OFF] = T(0);
}

// ---- ?potrf 'L' in place on the L blocks. Returns info (0, or k: leading minor k not positive definite). Collective.
constexpr int kInfoBase = 0x40000000;                      // *info_s = kInfoBase - (first bad pivot, 1-based), 0 = none

template <typename T, int NB>
__device__ __forceinline__ int lds_potrf(int n, T* smem, int* info_s, long long* dbg = nullptr)
{
    // MIR_LSQ_VARIANT_DEBUG_SOLVE: shader-clock cycles of thread 0 (wave 0 = the critical chain), summed over the panels:
    // dbg[20] diagonal update + factorisation, [21] wait at the barrier behind it, [22] rows below, [23] wait behind them
    long long ph[5] = {0, 0, 0, 0, 0};
#define MIRLSQ_PH(k_) do { if (dbg && threadIdx.x == 0) { const long long t_ = clock64(); ph[k_] += t_ - ph[4]; ph[4] = t_; } } while (0)
    using C = LdsSolveCfg<NB>;
    using Acc = typename Mma<T>::Acc;
    T* L = smem + C::L_OFF;
    T* rd = smem + C::RD_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *info_s = 0;
    const int nbl = (n + 15) >> 4;                           // blocks that hold real rows
    // ---- (1) the diagonal block k on wave 0, kept in the MFMA accumulator layout under the relabelling Mma::perm: lane
    //      (i = lane & 15, g = lane >> 4), register q holds A[row perm(i)][column 4 g + q] (the block is symmetric), so lane
    //      group g owns the four-column panel g. Panel jb: its group factors the four columns on the VALU (pivot and
    //      multipliers broadcast inside the 16-lane group: DPP), then every later column gets the rank-4 update as ONE
    //      v_mfma 16x16x4 whose k-slot t operand is panel column t -- moved from group jb to group t by three permlane swaps.
    //      The first version updated the 15 - c later columns of every pivot c one DPP broadcast + FMA at a time
    //      (120 x 28 cycles: 2.8 us per block; one MFMA per pivot: 3.4 us, the 16-pass MFMA latency sits on the chain).
    auto factor_diag = [&](int k, Acc acc) {
        const int i = lane & 15, g = lane >> 4;
        const int r = Mma<T>::perm(i);                       // the matrix row of this lane
        int bad = 0;
        // The IDENTITY rides along as a second block in the same layout (lane (i, g), register q: X[perm(i)][4 g + q]): the column
        // operations that turn A_kk into L_kk turn it into X = inv(L_kk)^T (A L^-T = L, I L^-T = L^-T; upper triangular, its
        // diagonal the reciprocal pivots). One more multiply / FMA per column operation and one more MFMA per four-column
        // panel on this wave's chain (+0.3 us a block) buy (a) the inverses of the diagonal blocks that ?potrs needs -- a pass
        // of their own before (2.5 us at n = 128) -- and (b) the rows below the diagonal block as a matrix product
        // L_Ik = A_Ik X on the matrix cores instead of one forward substitution per thread (step (2) below).
        Acc inv;
#pragma unroll
        for (int q = 0; q < 4; ++q) inv[q] = (r == 4 * g + q) ? T(1) : T(0);
        static_for<4>([&](auto jj) {
            constexpr int jb = decltype(jj)::value;
            if (g == jb) {
                static_for<4>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    constexpr int c = 4 * jb + t;
                    const T piv = dpp_row_bcast<Mma<T>::perm(c)>(acc[t]);
                    if (!(piv > 0)) { if (16 * k + c < n && bad == 0) bad = 16 * k + c + 1; }
                    T rinv, d;
                    rsqrt_sqrt(piv > 0 ? piv : T(1), rinv, d);
                    acc[t] = r > c ? acc[t] * rinv : (r == c ? d : acc[t]);
                    inv[t] = inv[t] * rinv;
                    if (r == c) rd[16 * k + c] = rinv;
                    static_for<4>([&](auto uu) {
                        constexpr int t2 = decltype(uu)::value;
                        if constexpr (t2 > t) {
                            const T l = dpp_row_bcast<Mma<T>::perm(4 * jb + t2)>(acc[t]);   // L[c2][c]
                            acc[t2] -= acc[t] * l;           // rows <= c: entries above the diagonal, which nothing reads
                            inv[t2] -= inv[t] * l;
                        }
                    });
                });
            }
            if constexpr (jb < 3) {
                const T x = group_pick<jb>(acc[0], acc[1], acc[2], acc[3]);      // lane (i, t) <- L[perm(i)][4 jb + t]
                const T xb = group_pick<jb>(inv[0], inv[1], inv[2], inv[3]);     // lane (i, t) <- X[perm(i)][4 jb + t]
                const T v = r > 4 * jb + 3 ? x : T(0);       // columns up to this panel are final: their operand is zero
                acc = Mma<T>::mma(-v, v, acc);               // A[r][c] -= sum_t L[r][4 jb + t] L[c][4 jb + t],  r, c > 4 jb + 3
                inv = Mma<T>::mma(-v, xb, inv);              // X[r][c] -= sum_t X[r][4 jb + t] L[c][4 jb + t],  c > 4 jb + 3
            }
        });
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * g + q;
            if (c <= r) L[blk_off(k, k, r, c)] = acc[q];
            else L[blk_off(k, k, r, c)] = inv[q];            // X(r, c) = inv(L_kk)(c, r), r < c: the unused upper triangle
        }
        // the first bad pivot (?potrf's info): groups see different pivots, keep the smallest index. Encoded so that 0 = none.
        if (bad != 0 && i == 0) atomicMax(info_s, kInfoBase - bad);
    };
    // block (I, I) in that layout / its update by panel k
    auto load_diag = [&](int I) {
        Acc acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = L[blk_off(I, I, Mma<T>::perm(lane & 15), 4 * (lane >> 4) + q)];
        return acc;
    };
    auto apply_panel_diag = [&](int k, int I, Acc acc) {     // acc -= L_Ik L_Ik^T (two MFMA chains: this is on wave 0's critical path)
        const int pr = Mma<T>::perm(lane & 15), lk = lane >> 4;
        T a[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) a[s4] = L[blk_off(I, k, pr, 4 * s4 + lk)];
        Acc acc2 = Acc{0, 0, 0, 0};
        acc = Mma<T>::mma(-a[0], a[0], acc);
        acc2 = Mma<T>::mma(-a[1], a[1], acc2);
        acc = Mma<T>::mma(-a[2], a[2], acc);
        acc2 = Mma<T>::mma(-a[3], a[3], acc2);
        return acc + acc2;
    };
    auto load_block = [&](int I, int J) {
        Acc acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = L[blk_off(I, J, Mma<T>::row(lane, q), lane & 15)];
        return acc;
    };
    auto apply_panel = [&](int k, int I, int J, Acc acc) {   // acc -= L_Ik L_Jk^T
        const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const T a = -L[blk_off(I, k, lr, 4 * s4 + lk)];
            const T b = L[blk_off(J, k, lr, 4 * s4 + lk)];
            acc = Mma<T>::mma(a, b, acc);
        }
        return acc;
    };
    // ---- (3) trailing update of one block on the matrix cores: block (I, J), k < J <= I, -= L_Ik L_Jk^T
    auto update_block = [&](int k, int I, int J) {
        const Acc acc = apply_panel(k, I, J, load_block(I, J));
#pragma unroll
        for (int q = 0; q < 4; ++q) L[blk_off(I, J, Mma<T>::row(lane, q), lane & 15)] = acc[q];
    };
    // Look-ahead: wave 0 brings block (k + 1, k + 1) up to date and factors it WHILE waves 1..3 apply panel k to the rest of
    // the trailing matrix -- the serial 16-pivot factorisation (~1.9 us) was a third of a panel step with three idle waves.
    __syncthreads();                                         // the blocks are loaded
    if (dbg && threadIdx.x == 0) ph[4] = clock64();
    if (wave == 0) factor_diag(0, load_diag(0));
    MIRLSQ_PH(0);
    for (int k = 0; k < nbl; ++k) {
        __syncthreads();                                     // block (k, k) is factored; column k carries every earlier update
        MIRLSQ_PH(1);
        if (*info_s != 0) return kInfoBase - *info_s;        // uniform
        if (k + 1 >= nbl) break;
        // ---- (2) rows below the diagonal block: L_Ik = A_Ik X, X = inv(L_kk)^T (upper triangle of block (k, k), its diagonal
        //      in rd), on the matrix cores: block I of wave w = (I - k - 1) mod 4, two blocks (two MFMA chains) a wave at a time.
        //      (Before: one forward substitution per row and thread, 120 dependent FMAs: 1.2 us of every panel step's 4.1.)
        {
            constexpr int roff = C::RD_OFF - C::L_OFF, zoff2 = C::ZERO_OFF - C::L_OFF;
            const int lr = lane & 15, lk = lane >> 4;
            T xb[4];                                          // B operand: X[4 s + lk][lr]
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int kk = 4 * s4 + lk;
                xb[s4] = L[kk < lr ? blk_off(k, k, kk, lr) : (kk == lr ? roff + 16 * k + kk : zoff2)];
            }
            for (int I0 = k + 1 + wave; I0 < nbl; I0 += 8) {
                const int I1 = I0 + 4;
                const bool two = I1 < nbl;
                T a0[4], a1[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    a0[s4] = L[blk_off(I0, k, lr, 4 * s4 + lk)];
                    a1[s4] = L[blk_off(two ? I1 : I0, k, lr, 4 * s4 + lk)];
                }
                Acc d0 = Acc{0, 0, 0, 0}, d1 = Acc{0, 0, 0, 0};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    d0 = Mma<T>::mma(a0[s4], xb[s4], d0);
                    d1 = Mma<T>::mma(a1[s4], xb[s4], d1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    L[blk_off(I0, k, Mma<T>::row(lane, q), lr)] = d0[q];
                    if (two) L[blk_off(I1, k, Mma<T>::row(lane, q), lr)] = d1[q];
                }
            }
        }
        MIRLSQ_PH(2);
        __syncthreads();
        MIRLSQ_PH(3);
        // ---- (3) + (1) of the next panel
        {
            const int rem = nbl - 1 - k, cnt = rem * (rem + 1) / 2;
            if (wave == 0) {
                const Acc d = apply_panel_diag(k, k + 1, load_diag(k + 1));
                factor_diag(k + 1, d);                       // block (k + 1, k + 1) goes from LDS through the registers once
            } else {
                // two blocks per step: the four MFMAs of a block are one dependent chain (64 issue cycles + the result wait
                // each), a second independent chain fills it
                constexpr int NW = kSolveThreads / kWave - 1;
                auto blk = [&](int idx, int& I, int& J) {
                    int ii = 0;
                    while ((ii + 1) * (ii + 2) / 2 <= idx) ++ii;
                    I = k + 1 + ii;
                    J = k + 1 + (idx - ii * (ii + 1) / 2);
                };
                for (int idx = wave; idx < cnt; idx += 2 * NW) {
                    int I0, J0, I1, J1;
                    blk(idx, I0, J0);
                    if (idx + NW < cnt) {
                        blk(idx + NW, I1, J1);
                        Acc a0 = load_block(I0, J0), a1 = load_block(I1, J1);
                        const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            a0 = Mma<T>::mma(-L[blk_off(I0, k, lr, 4 * s4 + lk)], L[blk_off(J0, k, lr, 4 * s4 + lk)], a0);
                            a1 = Mma<T>::mma(-L[blk_off(I1, k, lr, 4 * s4 + lk)], L[blk_off(J1, k, lr, 4 * s4 + lk)], a1);
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            L[blk_off(I0, J0, Mma<T>::row(lane, q), lr)] = a0[q];
                            L[blk_off(I1, J1, Mma<T>::row(lane, q), lr)] = a1[q];
                        }
                    } else {
                        update_block(k, I0, J0);
                    }
                }
            }
        }
        MIRLSQ_PH(0);
    }
    if (dbg && threadIdx.x == 0) { dbg[20] = ph[0]; dbg[21] = ph[1]; dbg[22] = ph[2]; dbg[23] = ph[3]; }
#undef MIRLSQ_PH
    // (the inverses of the diagonal blocks sit, transposed, in the unused upper triangles: factor_diag put them there)
    __syncthreads();
    return 0;
}

// ---- ?potrs on ONE wave, left-looking, no workgroup barrier inside: the vector sits in LDS and is solved in place block
//      by block. Lane (r = lane >> 2, h = lane & 3) works on row r of the current diagonal block and on every fourth column;
//      step j: z_j -= sum_{k < j} L_jk x_k (a DPP quad sum), then x_j = inv(L_jj) z_j through a 16-value exchange in LDS
//      (DS operations of one wave execute in order: wave_lds_fence for the compiler, no barrier). Software-pipelined and fully
//      unrolled: the products with x_0 .. x_{j-1} for step j + 1 are loaded before and summed WHILE step j's exchange is in
//      flight, so only the k = j - 1 term, two quad sums and two LDS write -> read exchanges (~160 cycles each, measured) are
//      on the chain of a step: ~550 cycles, 5.8 us at n = 128. Earlier versions: every row a thread with one workgroup
//      barrier + two LDS round trips per block step (9 us), then this lane layout with a run-time k loop (also 9 us: each k
//      iteration waited for its own loads). All four lanes of a quad store the (identical) results: no divergent branch.
//      Thread i < 16 NB passes z_i in and gets x_i back.
template <typename T, int NB>
__device__ __forceinline__ T lds_potrs(int n, T* smem, T z)
{
    using C = LdsSolveCfg<NB>;
    const T* L = smem + C::L_OFF;
    const T* rd = smem + C::RD_OFF;
    T* zv = smem + C::ZV_OFF;                                // the vector, solved in place
    T* xk = smem + C::XV_OFF;                                // 16-value exchange inside the wave
    const int tid = threadIdx.x;
    const int nbl = (n + 15) >> 4;
    constexpr int zoff = C::ZERO_OFF - C::L_OFF;             // an LDS element that holds 0
    if (tid < C::NV) zv[tid] = z;
    __syncthreads();
    if (tid < kWave) {
        const int r = tid >> 2, h = tid & 3;
        // one block step: x_j = D (z_j - pre - Lc . x_kc), D = inv(L_jj) (FWD) or its transpose; then the off-chain sum for the
        // next step. FWD: next = j + 1, products with blocks k <= j - 1 (x_j follows as that step's chain term).
        T pre = 0;
        static_for<NB>([&](auto jj) {                        // forward: L w = z
            constexpr int j = decltype(jj)::value;
            if (j < nbl) {
                T lc[4], zc[4];
                if constexpr (j > 0) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) { lc[cc] = L[blk_off(j, j - 1, r, h + 4 * cc)]; zc[cc] = zv[16 * (j - 1) + h + 4 * cc]; }
                }
                const T zj = zv[16 * j + r];
                T cf[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = h + 4 * cc;
                    const T t = L[c < r ? blk_off(j, j, c, r) : zoff];           // inv(r, c), c < r, stored at (c, r)
                    cf[cc] = t;
                }
                const T dg = rd[16 * j + r];
                // loads of the off-chain products of step j + 1
                T lo[j > 0 ? 4 * j : 1], zo[j > 0 ? 4 * j : 1];
                if constexpr (j > 0 && j + 1 < NB) {
                    static_for<j>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) { lo[4 * k + cc] = L[blk_off(j + 1, k, r, h + 4 * cc)]; zo[4 * k + cc] = zv[16 * k + h + 4 * cc]; }
                    });
                }
                T acc = pre;
                if constexpr (j > 0) {
                    T a1 = lc[1] * zc[1];
                    acc += lc[0] * zc[0];
                    a1 += lc[3] * zc[3];
                    acc += lc[2] * zc[2];
                    acc += a1;
                }
                acc = quad_sum(acc);
                const T zr = zj - acc;
                xk[r] = zr;                                  // the four lanes of a quad store the same value: no branch
                wave_lds_fence();
                T xs[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) xs[cc] = xk[h + 4 * cc];
                __builtin_amdgcn_sched_barrier(0);           // the exchange is in flight: this is where the off-chain sum belongs
                pre = 0;
                if constexpr (j > 0 && j + 1 < NB) {
                    T p1 = 0;
#pragma unroll
                    for (int e = 0; e < 4 * j; e += 2) { pre += lo[e] * zo[e]; p1 += lo[e + 1] * zo[e + 1]; }
                    pre += p1;
                }
                __builtin_amdgcn_sched_barrier(0);
                T s0 = 0, s1 = 0;
#pragma unroll
                for (int cc = 0; cc < 4; cc += 2) {
                    s0 += ((h + 4 * cc) == r ? dg : cf[cc]) * xs[cc];
                    s1 += ((h + 4 * cc + 4) == r ? dg : cf[cc + 1]) * xs[cc + 1];
                }
                const T s = quad_sum(s0 + s1);
                zv[16 * j + r] = s;
                wave_lds_fence();
            }
        });
        // backward: L^T x = w. Step j (from nbl - 1 down): chain term k = j + 1, off-chain k >= j + 2. Unrolled over
        // d = NB - 1 - j so that the compile-time index runs upwards; blocks >= nbl hold nothing and are skipped at run time.
        pre = 0;
        static_for<NB>([&](auto dd) {
            constexpr int j = NB - 1 - decltype(dd)::value;
            if (j < nbl) {
                const bool chain = j + 1 < nbl;
                T lc[4], zc[4];
                if constexpr (j + 1 < NB) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        lc[cc] = L[chain ? blk_off(j + 1, j, h + 4 * cc, r) : zoff];            // L(16 (j + 1) + c, 16 j + r)
                        zc[cc] = zv[16 * (j + 1) + h + 4 * cc];
                    }
                }
                const T zj = zv[16 * j + r];
                T cf[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = h + 4 * cc;
                    cf[cc] = L[c > r ? blk_off(j, j, r, c) : zoff];              // inv(c, r), c > r, stored at (r, c)
                }
                const T dg = rd[16 * j + r];
                constexpr int NO = (j > 0 && j + 1 < NB) ? NB - 1 - j : 0;       // off-chain blocks of step j - 1: k = j + 1 .. NB - 1
                T lo[NO > 0 ? 4 * NO : 1], zo[NO > 0 ? 4 * NO : 1];
                if constexpr (NO > 0) {
                    static_for<NO>([&](auto kk) {
                        constexpr int k = j + 1 + decltype(kk)::value;
                        const bool in = k < nbl;
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            lo[4 * (k - j - 1) + cc] = L[in ? blk_off(k, j - 1, h + 4 * cc, r) : zoff];
                            zo[4 * (k - j - 1) + cc] = zv[16 * k + h + 4 * cc];
                        }
                    });
                }
                T acc = pre;
                if constexpr (j + 1 < NB) {
                    T a1 = lc[1] * zc[1];
                    acc += lc[0] * zc[0];
                    a1 += lc[3] * zc[3];
                    acc += lc[2] * zc[2];
                    acc += a1;
                }
                acc = quad_sum(acc);
                const T zr = zj - acc;
                xk[r] = zr;                                  // the four lanes of a quad store the same value: no branch
                wave_lds_fence();
                T xs[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) xs[cc] = xk[h + 4 * cc];
                __builtin_amdgcn_sched_barrier(0);
                pre = 0;
                if constexpr (NO > 0) {
                    T p1 = 0;
#pragma unroll
                    for (int e = 0; e < 4 * NO; e += 2) { pre += lo[e] * zo[e]; p1 += lo[e + 1] * zo[e + 1]; }
                    pre += p1;
                }
                __builtin_amdgcn_sched_barrier(0);
                T s0 = 0, s1 = 0;
#pragma unroll
                for (int cc = 0; cc < 4; cc += 2) {
                    s0 += ((h + 4 * cc) == r ? dg : cf[cc]) * xs[cc];
                    s1 += ((h + 4 * cc + 4) == r ? dg : cf[cc + 1]) * xs[cc + 1];
                }
                const T s = quad_sum(s0 + s1);
                zv[16 * j + r] = s;
                wave_lds_fence();
            }
        });
    }
    __syncthreads();
    if (tid < C::NV) z = zv[tid];
    __syncthreads();                                         // zv / xv are free again
    return z;
}

// ---- r_i = b_i - (A x)_i, w_i = |b_i| + (|A| |x|)_i for the row of thread tid (valid for tid < n). x is published
//      through xv; two threads per row do the work, the owner thread of the row gets the result through zv. Collective.
template <typename T, int NB>
__device__ __forceinline__ void lds_residual(int n, T* smem, T bi, T xi, T& ri, T& wi)
{
    using C = LdsSolveCfg<NB>;
    const T* A = smem + C::A_OFF;
    T* xv = smem + C::XV_OFF;
    T* zv = smem + C::ZV_OFF;
    const int tid = threadIdx.x;
    const int nbl = (n + 15) >> 4;
    if (tid < C::NV) xv[tid] = tid < n ? xi : T(0);
    __syncthreads();
    T rs = 0, ws = 0;
    constexpr int PASSES = (2 * C::NV + kSolveThreads - 1) / kSolveThreads;
    T rk[PASSES], wk[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int i = (ps * kSolveThreads + tid) >> 1, h = tid & 1;
        const int I = i >> 4, r = i & 15;
        T ra = 0, wa = 0, ra1 = 0, wa1 = 0;                    // two chains each: 64 dependent FMAs were the critical path
        if (i < 16 * nbl) {
            for (int J = h; J < nbl; J += 2) {
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    const T a0 = J <= I ? A[blk_off(I, J, r, c)] : A[blk_off(J, I, c, r)];
                    const T a1 = J <= I ? A[blk_off(I, J, r, c + 1)] : A[blk_off(J, I, c + 1, r)];
                    const T x0 = xv[16 * J + c], x1 = xv[16 * J + c + 1];
                    ra += a0 * x0;
                    ra1 += a1 * x1;
                    wa += dabs(a0) * dabs(x0);
                    wa1 += dabs(a1) * dabs(x1);
                }
            }
        }
        ra += ra1;
        wa += wa1;
        ra += dpp_quad<0xB1>(ra);                              // the row's other half: the neighbouring lane (DPP, no LDS permute)
        wa += dpp_quad<0xB1>(wa);
        rk[ps] = ra; wk[ps] = wa;
    }
    __syncthreads();                                         // every read of xv is done: zv / xv may be rewritten
    // hand the row sums to the owner threads: row i's pair leader is thread 2 i (mod 256) of pass 2 i / 256
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int i = (ps * kSolveThreads + tid) >> 1;
        if ((tid & 1) == 0 && i < C::NV) { zv[i] = rk[ps]; xv[i] = wk[ps]; }
    }
    __syncthreads();
    if (tid < C::NV) { rs = zv[tid]; ws = xv[tid]; }
    __syncthreads();
    ri = bi - rs;
    wi = dabs(bi) + ws;
}

// ---------------------------------------------------------------- ?posvx('E','L'), nrhs = 1, in LDS
// src: n x n full symmetric in global memory (leading dimension ld); `shift` is added to its diagonal (the LM damping,
// LS:1079). Thread tid < n passes its right-hand-side entry bi and receives its solution entry in xi. Returns info.
template <typename T, int NB>
__device__ __forceinline__ int posvx_lds(int n, const T* src, int ld, T shift, T bi, T& xi, T* smem, T* red, int* info_s,
                                         long long* dbg, bool* scaled, LdsPreload<T, NB>& pre, bool preloaded)
{
    using C = LdsSolveCfg<NB>;
    const int tid = threadIdx.x;
    const T eps = Lim<T>::eps / 2;              // dlamch('Epsilon')
    const T safmin = Lim<T>::min_normal;        // dlamch('Safe minimum')
    T* zv = smem + C::ZV_OFF;
    MIRLSQ_STAMP(dbg, 2);
    if (!preloaded) lds_load_issue<T, NB>(n, src, ld, pre);     // (`pre` by reference, never through a pointer: the values stay in registers)
    // ?poequ / ?laqsy. The diagonal comes straight from the loaded values (the sum the LDS copy holds: t + shift), its minimum
    // and maximum in ONE workgroup reduction; the LDS blocks are written between the reduction's two barriers
    const T di = tid < n ? pre.diag + shift : T(0);
    if (dbg && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[24] = wall_clock64(); }   // every load has landed
    T smin, amax;
    {
        const T mn = wave_min(tid < n ? di : Lim<T>::inf()), mx = wave_max(tid < n ? di : -Lim<T>::inf());
        __syncthreads();                                       // red / the LDS blocks may still be read by the previous user
        if ((tid & 63) == 0) { red[tid >> 6] = mn; red[4 + (tid >> 6)] = mx; }
        lds_load_commit<T, NB>(n, pre, shift, smem);
        __syncthreads();
        smin = red[0]; amax = red[4];
#pragma unroll
        for (int w = 1; w < kSolveThreads / kWave; ++w) { smin = red[w] < smin ? red[w] : smin; amax = red[4 + w] > amax ? red[4 + w] : amax; }
    }
    MIRLSQ_STAMP(dbg, 25);
    bool rcequ = false;
    T si = 1;
    if (smin > 0) {
        const T scond = dsqrt(smin) / dsqrt(amax);
        si = tid < n ? T(1) / dsqrt(di) : T(1);
        const T small = safmin / Lim<T>::eps, large = T(1) / small;
        rcequ = !(scond >= T(0.1) && amax >= small && amax <= large);
    }
    if (scaled) *scaled = rcequ;
    if (rcequ) {                                             // uniform
        if (tid < C::NV) zv[tid] = si;
        __syncthreads();
        for (int e = tid; e < C::NBT * 256; e += kSolveThreads) {
            const int blk = e >> 8, r = e & 15, c = (e >> 4) & 15;
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= blk) ++I;
            const int J = blk - I * (I + 1) / 2;
            const T f = zv[16 * I + r] * zv[16 * J + c];
            const int o = blk * kLdsBlk + r + 17 * c;
            const T v = f * smem[C::A_OFF + o];
            smem[C::A_OFF + o] = v;
            smem[C::L_OFF + o] = v;
        }
        bi = si * bi;
        __syncthreads();
    }
    MIRLSQ_STAMP(dbg, 3);

    const int info = lds_potrf<T, NB>(n, smem, info_s, dbg);
    if (info != 0) return info;
    MIRLSQ_STAMP(dbg, 4);

    // ?potrs, then ?porfs (iterative refinement, ITMAX = 5): ONE call site of the triangular solves and of the residual -- round 0
    // solves for x, round c > 0 for the correction of the residual that round c - 1 left (two inlined copies of the unrolled
    // ?potrs were 15 KB of a 66 KB kernel: more than the instruction cache holds).
    // (Round 4 also built the residual of round c STREAMED under the backward sweep of its ?potrs -- waves 1, 2 one row each,
    // taking a block of x the moment wave 0 published it behind an LDS flag. Correct, and 0.25 us faster per round, not 1.6:
    // one thread a row needs 0.49 us per 16-column block where a sweep step takes 0.26, so the stream finished 1.75 us after
    // the sweep; the two-threads-a-row residual below does the same work in 2.0 us on four waves. Not kept.)
    const T safe1 = T(n + 1) * safmin, safe2 = safe1 / eps;
    T lstres = 3;
    T x = 0, zin = tid < n ? bi : T(0);
    for (int count = 0;; ++count) {
        const T dz = lds_potrs<T, NB>(n, smem, zin);
        x = count == 0 ? dz : x + dz;
        if (count == 0) MIRLSQ_STAMP(dbg, 5);
        if (count == 1) MIRLSQ_STAMP(dbg, 13);
        T ri, wi;
        lds_residual<T, NB>(n, smem, bi, x, ri, wi);
        if (count == 0) MIRLSQ_STAMP(dbg, 11);
        if (count == 1) MIRLSQ_STAMP(dbg, 14);
        T qv = 0;
        if (tid < n) qv = (wi > safe2) ? dabs(ri) / wi : (dabs(ri) + safe1) / (wi + safe1);
        const T berr = block_max(qv, red);
        if (count == 0) MIRLSQ_STAMP(dbg, 12);
        if (!(berr > eps && 2 * berr <= lstres && count + 1 <= 5)) break;
        lstres = berr;
        zin = tid < n ? ri : T(0);
    }
    xi = rcequ ? si * x : x;
    if (dbg && threadIdx.x == 0) dbg[15] = lstres == T(3) ? 0 : 1;     // was a correction applied?
    MIRLSQ_STAMP(dbg, 6);
    return 0;
}

}  // namespace mirlsq
