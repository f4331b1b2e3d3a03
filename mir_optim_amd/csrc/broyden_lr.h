// broyden_lr.h -- Broyden passes that do not rewrite J (gfx950).
//
// The reference's Broyden pass (least_squares.d:1002-1006, then 1052 and 1065) is
//     u = (y_new - y_old - J dx) / (dx.dx);  J += u dx^T;  Jy = J^T y_new;  JJ = J^T J
// i.e. three BLAS-2 sweeps plus a syrk over the updated m x n matrix. Here J stays what the last full
// refresh made it (J0) and the accepted updates are kept as k <= kLrMax pending rank-one terms,
//     J_k = J0 + U D^T          U: m x k (columns u_l),  D: n x k (columns dx_l),
// so that one READ-ONLY sweep over J0 (+ the k columns of U) yields everything a pass needs:
//     s_i   = J0[i,:].dx + sum_l U[i,l] (D_l.dx)                    (= (J_{k-1} dx)_i)
//     u_i   = -(1/dx.dx) ((y_old_i - y_new_i) + s_i)                (the reference's operation order)
//     v0    = J0^T u,  w_l = U_l.u,  uu = u.u                       -> v = J_{k-1}^T u = v0 + D w
//     g0    = J0^T y,  h_l = U_l.y,  uy = u.y                       -> J_k^T y = g0 + D h + dx uy
//     J_k^T J_k = J_{k-1}^T J_{k-1} + v dx^T + dx v^T + uu dx dx^T  (n x n, k_lr_finish)
// HBM traffic per pass: T (m n + (k + 3) m) instead of T (2 m n + 3 m); no n^2 work per row at all.
// When k reaches the cap the pending terms are folded into J (k_lr_flush: J += sum_l u_l dx_l^T, applied in
// update order like the reference's successive `ger`s).
//
// Lane map of the sweep (same as the residual kernels): a wave takes 4 rows per step, lane (q = lane >> 4,
// p = lane & 15) reads the column pairs [32 c + 2 p, + 1] of row 4 g + q, so every 16-lane group streams 256
// contiguous bytes per load. Lane p of a group also owns pending column l = p: it carries U[i, p] into the
// row dot product and accumulates w_p and h_p (hence kLrMax = 16). Each wave walks a contiguous range of
// rows, per-block partials are summed in a fixed order by k_lr_reduce: results are bitwise reproducible.
#pragma once

#include "broyden_launch.h"
#include "common.h"
#include "solve_types.h"

namespace mirlsq {


template <typename T> struct LrPair;
template <> struct LrPair<double> { using type = double2; };
template <> struct LrPair<float> { using type = float2; };



// May a Broyden pass follow the trial of ladder record r if it is accepted? (The speculative sweep and the decision of the
// fused round evaluate the same expression on the same record.) No after a QP failure, a NaN, the step guard, a null step
// or the gradient test (LS:1053-1106), and no when the x-convergence test LS:1164-1173 fires: it forces a full refresh.
template <typename T>
__device__ __forceinline__ bool lr_spec_go(const ChainRec<T>& r, T absTolerance, T relTolerance)
{
    const T dxn = dsqrt(r.new_dx_dot);
    return r.qp_status == 0 && !(r.flags & (kFlagDxNaN | kFlagXNaN | kFlagStepTooLong | kFlagNullStep | kFlagGradSmall))
        && (dxn > absTolerance && r.trial_xnorm > dxn * relTolerance);
}

// ||y||^2 over the 4-row groups [gb, ge) of one wave, lane group q taking row 4 g + q: every lane of the group forms the same
// sum, in row order, one fma per row; the caller adds the four groups (xor 16, xor 32). Rows past m count as zero.
template <typename T>
__device__ __forceinline__ T lr_wave_sumsq(const T* __restrict__ y, size_t m, size_t gb, size_t ge, int q)
{
    T yy = 0;
    for (size_t g0 = gb; g0 < ge; g0 += 8) {
        T v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const size_t row = 4 * (g0 + e) + q;
            v[e] = (g0 + e < ge && row < m) ? y[row] : T(0);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (g0 + e < ge) yy = dfma(v[e], v[e], yy);
    }
    yy += wave_shfl_xor(yy, 16);
    yy += wave_shfl_xor(yy, 32);
    return yy;
}

// (four workgroups per CU up to n = 128)
template <typename T, int NCP, bool VEC>
__global__ __launch_bounds__(256, (NCP <= 4 ? 4 : 1)) void k_broyden_lr(const LrArgs<T> a)
{
    using P2 = typename LrPair<T>::type;
    __shared__ T red[4][lr_len(kLrMaxN)];
    __shared__ T coef[kLrMax];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, p = lane & 15;
    const int n = a.n, k = a.k;
    const size_t m = a.m;
    // the rows of this wave: a contiguous range of 4-row groups (k_lr_sumsq walks the same ranges)
    const size_t G = (m + 3) / 4;
    const size_t nw = (size_t)gridDim.x * 4, wid = (size_t)blockIdx.x * 4 + wave;
    const size_t per = (G + nw - 1) / nw;
    const size_t gb = wid * per;
    const size_t ge = gb + per < G ? gb + per : G;
    if (a.spec_rec && !lr_spec_go(a.spec_rec[0], a.absTolerance, a.relTolerance)) {
        // speculative sweep whose trial cannot be followed by a Broyden pass: ||y||^2 only, zeros for the rest
        const T yy = lr_wave_sumsq(a.y, m, gb, ge, q);
        if (q == 0 && p == 0) red[wave][0] = yy;
        __syncthreads();
        const int len = lr_len(n);
        T* out = a.partials + (size_t)blockIdx.x * len;
        for (int e = threadIdx.x; e < len; e += blockDim.x) out[e] = e == lr_yy(n) ? (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]) : T(0);
        return;
    }

    // coef[l] = D_l . dx for the pending columns (wave w takes l = w, w + 4, ...)
    for (int l = wave; l < k; l += 4) {
        T s = 0;
        for (int j = lane; j < n; j += kWave) s += a.D[(size_t)l * n + j] * a.dx[j];
        s = wave_sum(s);
        if (lane == 0) coef[l] = s;
    }
    T d0[NCP], d1[NCP];
    int coff[NCP];
    bool ok0[NCP], ok1[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
        const int col = 32 * c + 2 * p;
        ok0[c] = col < n;
        ok1[c] = col + 1 < n;
        coff[c] = ok0[c] ? col : 0;
        const T t0 = a.dx[ok0[c] ? col : 0], t1 = a.dx[ok1[c] ? col + 1 : 0];
        d0[c] = ok0[c] ? t0 : T(0);
        d1[c] = ok1[c] ? t1 : T(0);
    }
    __syncthreads();
    const bool own = p < k;
    const T cp = own ? coef[p] : T(0);
    const T* __restrict__ Up = a.U + (size_t)(own ? p : 0) * m;
    T* __restrict__ Uk = a.U + (size_t)k * m;
    const T nd = -(T(1) / *a.dx_dot);

    T va0[NCP], va1[NCP], ga0[NCP], ga1[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) { va0[c] = 0; va1[c] = 0; ga0[c] = 0; ga1[c] = 0; }
    T wl = 0, hl = 0, uu = 0, uy = 0, yy = 0;

    // UNR row groups per trip: their loads are issued together, then the groups are taken in order (the same per-lane order of
    // sums as one group per trip: same bits). Narrow problems (n <= 64: one or two 16-byte loads a lane and group) are small
    // in bytes as well -- cfg 2's J is 12.8 MB, 8 groups a wave -- and ran at one memory latency PER GROUP (12 us for a 2 us
    // sweep); n = 128 and above keep one group per trip (HBM-bound at 0.73-0.78 of the peak, four workgroups a CU by registers).
    constexpr int UNR = NCP == 1 ? 4 : (NCP == 2 ? 2 : 1);
    for (size_t g0 = gb; g0 < ge; g0 += UNR) {
        T v0[UNR][NCP], v1[UNR][NCP], ynv[UNR], yov[UNR], ulv[UNR];
#pragma unroll
        for (int e = 0; e < UNR; ++e) {
            const size_t g = g0 + e < ge ? g0 + e : ge - 1;    // (a trip's spare slots re-read the last group: valid addresses)
            const size_t row = 4 * g + q;
            const size_t rr = row < m ? row : m - 1;
            const T* __restrict__ rp = a.J + rr * (size_t)n;
#pragma unroll
            for (int c = 0; c < NCP; ++c) {
                if constexpr (VEC) {
                    typedef T lr_v2 __attribute__((ext_vector_type(2)));
                    const lr_v2 t = __builtin_nontemporal_load(reinterpret_cast<const lr_v2*>(rp + coff[c]));   // J is swept once per pass and is larger than the Infinity Cache
                    v0[e][c] = t.x;
                    v1[e][c] = t.y;
                } else {
                    v0[e][c] = rp[coff[c]];
                    v1[e][c] = rp[ok1[c] ? coff[c] + 1 : 0];
                }
            }
            ynv[e] = a.y[rr];
            yov[e] = a.y_old[rr];
            ulv[e] = own ? Up[rr] : T(0);
        }
#pragma unroll
        for (int e = 0; e < UNR; ++e) {
            if (g0 + e < ge) {
                const size_t row = 4 * (g0 + e) + q;
                const bool rok = row < m;
                T yn = ynv[e];
                const T yo = yov[e], ul = ulv[e];
                T s = ul * cp;
#pragma unroll
                for (int c = 0; c < NCP; ++c) s += v0[e][c] * d0[c] + v1[e][c] * d1[c];
                s = sum16(s);
                T u = nd * ((yo - yn) + s);                        // LS:1003-1005: axpy(-1, y, mBuffer); gemv; scal(-d)
                if (!rok) { u = 0; yn = 0; }
                if (rok && p == 0) Uk[row] = u;
#pragma unroll
                for (int c = 0; c < NCP; ++c) {
                    va0[c] += v0[e][c] * u;
                    va1[c] += v1[e][c] * u;
                    ga0[c] += v0[e][c] * yn;
                    ga1[c] += v1[e][c] * yn;
                }
                wl += ul * u;
                hl += ul * yn;
                uu += u * u;
                uy += u * yn;
                yy = dfma(yn, yn, yy);                             // (the operation and the order of lr_wave_sumsq)
            }
        }
    }

    // the four row groups of the wave, then the four waves of the block, in a fixed order
    auto qsum = [](T v) { v += wave_shfl_xor(v, 16); v += wave_shfl_xor(v, 32); return v; };
#pragma unroll
    for (int c = 0; c < NCP; ++c) { va0[c] = qsum(va0[c]); va1[c] = qsum(va1[c]); ga0[c] = qsum(ga0[c]); ga1[c] = qsum(ga1[c]); }
    wl = qsum(wl); hl = qsum(hl); uu = qsum(uu); uy = qsum(uy); yy = qsum(yy);
    if (q == 0) {
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            if (ok0[c]) { red[wave][coff[c]] = va0[c]; red[wave][n + coff[c]] = ga0[c]; }
            if (ok1[c]) { red[wave][coff[c] + 1] = va1[c]; red[wave][n + coff[c] + 1] = ga1[c]; }
        }
        red[wave][2 * n + p] = wl;
        red[wave][2 * n + kLrMax + p] = hl;
        if (p == 0) { red[wave][2 * n + 2 * kLrMax] = uu; red[wave][2 * n + 2 * kLrMax + 1] = uy; red[wave][lr_yy(n)] = yy; }
    }
    __syncthreads();
    const int len = lr_len(n);
    T* out = a.partials + (size_t)blockIdx.x * len;
    for (int e = threadIdx.x; e < len; e += blockDim.x) out[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// sum the per-block partial vectors in a fixed order: blockDim = 1024 = 32 entries x 32 block ranges (a thread walks
// nparts / 32 partials with 8 loads in flight: the 8-range version spent 9 us on 16 dependent L2 round trips per thread)
template <typename T>
__global__ __launch_bounds__(32 * kReduceRanges) void k_lr_reduce(const T* __restrict__ partials, int nparts, int len, T* __restrict__ out)
{
    __shared__ T part[kReduceRanges][33];
    const int es = threadIdx.x & 31, sp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + es;
    T s = 0;
    if (e < len) {
        const int per = (nparts + kReduceRanges - 1) / kReduceRanges;
        const int b0 = sp * per, b1 = (b0 + per < nparts) ? b0 + per : nparts;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += partials[(size_t)b * len + e];
    }
    part[sp][es] = s;
    __syncthreads();
    if (sp == 0 && e < len) {
        T tot = part[0][es];
#pragma unroll
        for (int r = 1; r < kReduceRanges; ++r) tot += part[r][es];
        out[e] = tot;
    }
}

// the same sum for ONE entry (stride 1 between the partials), by a workgroup of any size >= kReduceRanges threads: ranges in
// sequence, then the range totals in sequence -- the bits k_lr_reduce gives that entry. Collective; every thread gets the sum.
template <typename T>
__device__ inline T lr_reduce_scalar(const T* __restrict__ partials, int nparts, T* part /* kReduceRanges, LDS */)
{
    if ((int)threadIdx.x < kReduceRanges) {
        const int per = (nparts + kReduceRanges - 1) / kReduceRanges;
        const int b0 = (int)threadIdx.x * per, b1 = (b0 + per < nparts) ? b0 + per : nparts;
        T s = 0;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += partials[b];
        part[threadIdx.x] = s;
    }
    __syncthreads();
    T tot = part[0];
#pragma unroll
    for (int r = 1; r < kReduceRanges; ++r) tot += part[r];
    __syncthreads();                                         // part may be reused
    return tot;
}

// ||v_k||^2, k = blockIdx.y, without a sweep: the row walk of k_broyden_lr over the vector alone (lr_wave_sumsq), one partial per
// workgroup at partials + k pstride + blockIdx.x -- summed by lr_reduce_scalar (k_decide_chain on one GPU, k_lr_sumsq_final
// before an all-reduce) the trial's sum of squares has the bits of entry lr_yy(n) of a reduced sweep
template <typename T>
__global__ __launch_bounds__(256) void k_lr_sumsq(const T* __restrict__ v0, size_t m, size_t vstride, T* __restrict__ partials0, int pstride)
{
    __shared__ T red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* __restrict__ v = v0 + (size_t)blockIdx.y * vstride;
    const size_t G = (m + 3) / 4;
    const size_t nw = (size_t)gridDim.x * 4, wid = (size_t)blockIdx.x * 4 + wave;
    const size_t per = (G + nw - 1) / nw;
    const size_t gb = wid * per;
    const size_t ge = gb + per < G ? gb + per : G;
    const T yy = lr_wave_sumsq(v, m, gb, ge, lane >> 4);
    if (lane == 0) red[wave] = yy;
    __syncthreads();
    if (threadIdx.x == 0) partials0[(size_t)blockIdx.y * pstride + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
template <typename T>
__global__ __launch_bounds__(256) void k_lr_sumsq_final(const T* __restrict__ partials0, int pstride, int nparts, T* __restrict__ out)
{
    __shared__ T part[kReduceRanges];
    const T tot = lr_reduce_scalar(partials0 + (size_t)blockIdx.x * pstride, nparts, part);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// n x n side of a pass: JJ += v dx^T + dx v^T + uu dx dx^T (block i < n: row i), Jy and |Jy|_inf (LS:1052-1053),
// D_k = dx (block n). `lr` is the (all-reduced) vector of k_lr_reduce.
template <typename T>
__global__ __launch_bounds__(256) void k_lr_finish(const T* __restrict__ lr, T* __restrict__ D, const T* __restrict__ dx,
                                                   int k, int n, T* __restrict__ JJ, T* __restrict__ Jy, LmState<T>* st)
{
    __shared__ T v[kLrMaxN];
    __shared__ T red[4];
    const T* __restrict__ w = lr + 2 * n;
    const T* __restrict__ h = w + kLrMax;
    const T uu = lr[2 * n + 2 * kLrMax], uy = lr[2 * n + 2 * kLrMax + 1];
    const int i = blockIdx.x;
    if (i < n) {
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            T s = lr[j];
            for (int l = 0; l < k; ++l) s = dfma(D[(size_t)l * n + j], w[l], s);     // (explicit: the fused round's kernel forms the same sums)
            v[j] = s;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const int r = i >= j ? i : j, c = i >= j ? j : i;      // one expression for (i, j) and (j, i): exactly symmetric
            JJ[(size_t)i * n + j] += lr_jj_term(v[r], v[c], dx[r], dx[c], uu);
        }
        return;
    }
    T mx = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        T s = lr[n + j];
        for (int l = 0; l < k; ++l) s = dfma(D[(size_t)l * n + j], h[l], s);
        s = dfma(dx[j], uy, s);
        Jy[j] = s;
        const T av = dabs(s);
        if (av > mx) mx = av;
        D[(size_t)k * n + j] = dx[j];
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        T r = red[0];
        for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) r = red[wv] > r ? red[wv] : r;
        st->jy_inf = r;
    }
}

// fold the k pending terms into J: J[i,:] += u_0[i] dx_0 + ... + u_{k-1}[i] dx_{k-1}, in update order
template <typename T, int NCP, bool VEC>
__global__ __launch_bounds__(256) void k_lr_flush(T* J, const T* __restrict__ U, const T* __restrict__ D,
                                                  int k, size_t m, int n)
{
    using P2 = typename LrPair<T>::type;
    extern __shared__ unsigned char lr_smem[];
    T* Ds = reinterpret_cast<T*>(lr_smem);                 // k x n
    for (int e = threadIdx.x; e < k * n; e += blockDim.x) Ds[e] = D[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    int coff[NCP];
    bool ok0[NCP], ok1[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
        const int col = 32 * c + 2 * p;
        ok0[c] = col < n;
        ok1[c] = col + 1 < n;
        coff[c] = ok0[c] ? col : 0;
    }
    const size_t G = (m + 3) / 4;
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t g = wave_id; g < G; g += nwaves) {
        const size_t row = 4 * g + q;
        const bool rok = row < m;
        const size_t rr = rok ? row : m - 1;
        T* rp = J + rr * (size_t)n;
        const T* sp = rp;
        T v0[NCP], v1[NCP];
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            if constexpr (VEC) {
                typedef T lr_v2 __attribute__((ext_vector_type(2)));
                const lr_v2 t = __builtin_nontemporal_load(reinterpret_cast<const lr_v2*>(sp + coff[c]));
                v0[c] = t.x;
                v1[c] = t.y;
            } else {
                v0[c] = sp[coff[c]];
                v1[c] = sp[ok1[c] ? coff[c] + 1 : 0];
            }
        }
        for (int l = 0; l < k; ++l) {
            const T ul = U[(size_t)l * m + rr];
#pragma unroll
            for (int c = 0; c < NCP; ++c) {
                v0[c] += ul * Ds[l * n + coff[c]];
                v1[c] += ul * Ds[l * n + (ok1[c] ? coff[c] + 1 : 0)];
            }
        }
        if (rok) {
#pragma unroll
            for (int c = 0; c < NCP; ++c) {
                if constexpr (VEC) {
                    if (ok0[c]) {
                        typedef T lr_v2 __attribute__((ext_vector_type(2)));
                        lr_v2 t; t.x = v0[c]; t.y = v1[c];
                        __builtin_nontemporal_store(t, reinterpret_cast<lr_v2*>(rp + coff[c]));
                    }
                } else {
                    if (ok0[c]) rp[coff[c]] = v0[c];
                    if (ok1[c]) rp[coff[c] + 1] = v1[c];
                }
            }
        }
    }
}

}  // namespace mirlsq
