// solve_lds.h -- ?posvx('E','L'), one right-hand side, entirely in LDS for n <= 128 (gfx950): the latency path of k_lm_solve.
//
// Replaces mir-lapack's posvx as the reference calls it (boxcqp.d:194, 310; Netlib ?posvx = ?poequ + ?laqsy + ?potrf +
// ?potrs + ?porfs; ?pocon and the forward-error bound are not computed: they only feed rcond / ferr, which the reference
// ignores, boxcqp.d:212, 323). The first version of this path (potrf_tiled2 / potrs_blocked, round 1) took ~150 us at
// n = 128 -- 13 % of a cfg-3 solve on one CU and most of a strong-scaled one: a column-by-column Cholesky (1 100 cycles
// per column), one-wave triangular solves built on v_readlane broadcasts, and refinement mat-vecs that fetched the matrix
// from L2 one dependent round trip at a time. This version keeps BOTH the factor and the matrix in LDS:
//
//   * storage: the lower BLOCK triangle of 16 x 16 blocks, block (I, J), J <= I, at ((I (I + 1) / 2 + J) * 272), element
//     (r, c) at r + 17 c. The leading dimension 17 makes row walks, column walks and the MFMA operand pattern
//     (r = lane & 15, c = 4 s + (lane >> 4)) all bank-conflict-free. n = 128: 36 blocks = 78 KB per matrix, 157 KB for
//     L and A together plus 3 KB of vectors: fits the 160 KB of a CU's LDS;
//   * ?potrf: right-looking by 16-column panels, three barriers per panel: (1) wave 0 factors the diagonal block in
//     registers (16 pivots, DPP row broadcasts, rsqrt + Newton like potrf_panel), (2) one thread per row below solves
//     its 16 panel entries against it, (3) the trailing update A_IJ -= L_Ik L_Jk^T runs on the matrix cores
//     (v_mfma_f64_16x16x4: four per block, operands and accumulators straight from / to the LDS blocks);
//   * the inverses of the diagonal blocks (one thread per column) are stored TRANSPOSED IN THE UNUSED UPPER TRIANGLES
//     of the diagonal blocks of L (their diagonals are the reciprocal pivots, kept in a vector): no extra storage;
//   * ?potrs: one thread per row, the vector in registers, ONE barrier per block step: the 16 owners of a diagonal block
//     are a DPP row -- they form x_k = inv(L_kk) z_k with row broadcasts (row_newbcast: no LDS exchange, no readlane),
//     publish it, and after the barrier every remaining row subtracts its 16 products from LDS;
//   * ?porfs: mat-vec with two threads per row from the LDS copy of A (equilibrated when ?laqsy says so), same berr
//     test and ITMAX = 5 as Netlib.
// Rows and columns past n are an identity extension, so no routine needs row masks.
#pragma once

#include "common.h"

namespace mirlsq {

constexpr int kLdsBlk = 272;     // 16 x 17 elements per block

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
__host__ __device__ constexpr int lds_solve_elems(int nb) { return nb * (nb + 1) * kLdsBlk + 48 * nb + 2; }

__device__ __forceinline__ int blk_off(int I, int J, int r, int c) { return (I * (I + 1) / 2 + J) * kLdsBlk + r + 17 * c; }

// ---- load: L = A = src (+ shift on the diagonal), identity past n. src: n x n full symmetric, leading dimension ld.
template <typename T, int NB>
__device__ __forceinline__ void lds_load_blocks(int n, const T* __restrict__ src, int ld, T shift, T* smem)
{
    using C = LdsSolveCfg<NB>;
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;      // lanes along a row of src: coalesced reads
    T v[C::NBT];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            const int gi = 16 * I + r, gj = 16 * J + c;
            const bool in = gi < n && gj < n;
            const T t = src[(size_t)(in ? gi : 0) * ld + (in ? gj : 0)];
            v[I * (I + 1) / 2 + J] = in ? (gi == gj ? t + shift : t) : (gi == gj ? T(1) : T(0));
        }
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
            const int o = blk_off(I, J, r, c);
            smem[C::L_OFF + o] = v[I * (I + 1) / 2 + J];
            smem[C::A_OFF + o] = v[I * (I + 1) / 2 + J];
        }
    if (threadIdx.x == 0) smem[C::ZERO_OFF] = T(0);
}

// ---- ?potrf 'L' in place on the L blocks. Returns info (0, or k: leading minor k not positive definite). Collective.
template <typename T, int NB>
__device__ __forceinline__ int lds_potrf(int n, T* smem, int* info_s)
{
    using C = LdsSolveCfg<NB>;
    using Acc = typename Mma<T>::Acc;
    T* L = smem + C::L_OFF;
    T* rd = smem + C::RD_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *info_s = 0;
    const int nbl = (n + 15) >> 4;                           // blocks that hold real rows
    for (int k = 0; k < nbl; ++k) {
        __syncthreads();                                     // block (k, k) carries every earlier panel's update
        // ---- (1) the diagonal block, wave 0, row per lane (lanes 16..63 shadow rows 0..15 and write nothing)
        if (wave == 0) {
            const int r = lane & 15;
            T p[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) p[c] = L[blk_off(k, k, r, c)];
            int bad = 0;
            // pivot c: every lane of the row group gets lane c's / lane c2's value with a DPP row broadcast (a VALU move:
            // no v_readlane -> SGPR -> VALU round trip, whose wait states dominated the first version of this loop)
            static_for<16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const T piv = dpp_row_bcast<c>(p[c]);
                if (!(piv > 0)) { if (16 * k + c < n && bad == 0) bad = 16 * k + c + 1; }
                T rinv, d;
                rsqrt_sqrt(piv > 0 ? piv : T(1), rinv, d);
                if (r > c) p[c] *= rinv; else if (r == c) p[c] = d;
                if (lane == c) rd[16 * k + c] = rinv;
                static_for<16>([&](auto cc2) {
                    constexpr int c2 = decltype(cc2)::value;
                    if constexpr (c2 > c) {
                        // rows r <= c only touch entries above the diagonal here (c2 > c >= r), which nothing reads: no mask
                        const T lc = dpp_row_bcast<c2>(p[c]);    // L[c2][c]
                        p[c2] -= p[c] * lc;
                    }
                });
            });
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) if (c <= r) L[blk_off(k, k, r, c)] = p[c];
            }
            if (bad != 0 && lane == 0) *info_s = bad;
        }
        __syncthreads();
        if (*info_s != 0) return *info_s;                    // uniform
        if (k + 1 >= nbl) break;
        // ---- (2) rows below the diagonal block against L_kk: one thread per row
        if (tid < 16 * (nbl - 1 - k)) {
            const int i = 16 * (k + 1) + tid, I = i >> 4, r = i & 15;
            T p[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) p[c] = L[blk_off(I, k, r, c)];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                T v = p[c];
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t < c) v -= p[t] * L[blk_off(k, k, c, t)];       // same address in every lane: LDS broadcast
                p[c] = v * rd[16 * k + c];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) L[blk_off(I, k, r, c)] = p[c];
        }
        __syncthreads();
        // ---- (3) trailing update on the matrix cores: block (I, J), k < J <= I, -= L_Ik L_Jk^T
        {
            const int rem = nbl - 1 - k, cnt = rem * (rem + 1) / 2;
            const int lr = lane & 15, lk = lane >> 4;
            for (int idx = wave; idx < cnt; idx += kSolveThreads / kWave) {
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= idx) ++ii;
                const int I = k + 1 + ii, J = k + 1 + (idx - ii * (ii + 1) / 2);
                Acc acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = L[blk_off(I, J, Mma<T>::row(lane, q), lr)];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const T a = -L[blk_off(I, k, lr, 4 * s4 + lk)];
                    const T b = L[blk_off(J, k, lr, 4 * s4 + lk)];
                    acc = Mma<T>::mma(a, b, acc);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) L[blk_off(I, J, Mma<T>::row(lane, q), lr)] = acc[q];
            }
        }
    }
    __syncthreads();
    // ---- inverses of the diagonal blocks: thread (k, c) forms column c of inv(L_kk) by forward substitution and stores
    //      its strictly-lower entries inv(r, c), r > c, TRANSPOSED at (c, r) -- the unused upper triangle of block (k, k)
    if (tid < 16 * nbl) {
        const int k = tid >> 4, c = tid & 15;
        T x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            T s = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < r) s -= L[blk_off(k, k, r, q)] * x[q];
            x[r] = r < c ? T(0) : (r == c ? rd[16 * k + r] : s * rd[16 * k + r]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) if (r > c) L[blk_off(k, k, c, r)] = x[r];
    }
    __syncthreads();
    return 0;
}

// ---- ?potrs: thread i < 16 NB holds z_i (the right-hand side on entry, the solution on return). Collective.
template <typename T, int NB>
__device__ __forceinline__ void lds_potrs(int n, T* smem, T& z)
{
    using C = LdsSolveCfg<NB>;
    const T* L = smem + C::L_OFF;
    const T* rd = smem + C::RD_OFF;
    T* xv = smem + C::XV_OFF;
    const int tid = threadIdx.x, I = tid >> 4, r = tid & 15;
    const int nbl = (n + 15) >> 4;
    const bool row = tid < 16 * nbl;
    constexpr int zoff = C::ZERO_OFF - C::L_OFF;             // an LDS element that holds 0
    // forward: L w = z
    for (int k = 0; k < nbl; ++k) {
        if (row && I == k) {
            // x_k = inv(L_kk) z_k inside the DPP row of the 16 owners: the coefficients inv(r, c), c < r, sit at (c, r) of the
            // diagonal block (addresses known up front: the 16 LDS reads are independent), z_c arrives by row broadcast
            T acc[4] = {rd[16 * k + r] * z, 0, 0, 0};        // inv(r, r) z_r
            static_for<16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                // c >= r: the coefficient is 0 -- read the zero kept at the end of the vectors instead of selecting a double
                const T cf = L[c < r ? blk_off(k, k, c, r) : zoff];
                const T zc = dpp_row_bcast<c>(z);
                acc[c & 3] += cf * zc;
            });
            z = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            xv[16 * k + r] = z;
        }
        __syncthreads();
        if (row && I > k) {
            T acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c & 3] += L[blk_off(I, k, r, c)] * xv[16 * k + c];
            z -= (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
    }
    // backward: L^T x = w
    for (int k = nbl - 1; k >= 0; --k) {
        if (row && I == k) {
            T acc[4] = {rd[16 * k + r] * z, 0, 0, 0};
            static_for<16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const T cf = L[c > r ? blk_off(k, k, r, c) : zoff];   // inv(c, r), c > r, stored at (r, c)
                const T zc = dpp_row_bcast<c>(z);
                acc[c & 3] += cf * zc;
            });
            z = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            xv[16 * k + r] = z;
        }
        __syncthreads();
        if (row && I < k) {
            T acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c & 3] += L[blk_off(k, I, c, r)] * xv[16 * k + c];   // L(16 k + c, 16 I + r)
            z -= (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
    }
    __syncthreads();                                         // xv is free again
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
        T ra = 0, wa = 0;
        if (i < 16 * nbl) {
            for (int J = h; J < nbl; J += 2) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const T a = J <= I ? A[blk_off(I, J, r, c)] : A[blk_off(J, I, c, r)];
                    const T xc = xv[16 * J + c];
                    ra += a * xc;
                    wa += dabs(a) * dabs(xc);
                }
            }
        }
        ra += wave_shfl_xor(ra, 1);
        wa += wave_shfl_xor(wa, 1);
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
                                         long long* dbg = nullptr, bool* scaled = nullptr)
{
    using C = LdsSolveCfg<NB>;
    const int tid = threadIdx.x;
    const T eps = Lim<T>::eps / 2;              // dlamch('Epsilon')
    const T safmin = Lim<T>::min_normal;        // dlamch('Safe minimum')
    T* zv = smem + C::ZV_OFF;
    MIRLSQ_STAMP(dbg, 2);
    lds_load_blocks<T, NB>(n, src, ld, shift, smem);
    __syncthreads();

    // ?poequ / ?laqsy
    const T di = tid < n ? smem[C::A_OFF + blk_off(tid >> 4, tid >> 4, tid & 15, tid & 15)] : T(0);
    const T smin = block_min(tid < n ? di : Lim<T>::inf(), red);
    const T amax = block_max(tid < n ? di : -Lim<T>::inf(), red);
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

    const int info = lds_potrf<T, NB>(n, smem, info_s);
    if (info != 0) return info;
    MIRLSQ_STAMP(dbg, 4);

    T x = tid < n ? bi : T(0);
    lds_potrs<T, NB>(n, smem, x);
    MIRLSQ_STAMP(dbg, 5);

    // ?porfs: iterative refinement, ITMAX = 5
    const T safe1 = T(n + 1) * safmin, safe2 = safe1 / eps;
    T lstres = 3;
    for (int count = 1;; ++count) {
        T ri, wi;
        lds_residual<T, NB>(n, smem, bi, x, ri, wi);
        if (count == 1) MIRLSQ_STAMP(dbg, 11);
        T qv = 0;
        if (tid < n) qv = (wi > safe2) ? dabs(ri) / wi : (dabs(ri) + safe1) / (wi + safe1);
        const T berr = block_max(qv, red);
        if (count == 1) MIRLSQ_STAMP(dbg, 12);
        if (berr > eps && 2 * berr <= lstres && count <= 5) {
            T dz = tid < n ? ri : T(0);
            lds_potrs<T, NB>(n, smem, dz);
            x += dz;
            lstres = berr;
            continue;
        }
        break;
    }
    xi = rcequ ? si * x : x;
    MIRLSQ_STAMP(dbg, 6);
    return 0;
}

}  // namespace mirlsq
