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

#include "common.h"
#include "solve_kernel.h"

namespace mirlsq {


template <typename T> struct LrPair;
template <> struct LrPair<double> { using type = double2; };
template <> struct LrPair<float> { using type = float2; };

template <typename T>
struct LrArgs {
    const T* J;        // m x n row-major, read only (J0)
    T* U;              // kLrMax columns of length m (column l at U + l m); column k is written
    const T* D;        // kLrMax x n, rows 0..k-1 valid
    const T* dx;       // the accepted step of this update (length n)
    const T* dx_dot;   // device scalar dx.dx (LS:1002: d = 1 / deltaX_dot)
    const T* y;        // residual at the new point
    const T* y_old;    // residual at the previous point (the reference's mBuffer after the swap LS:1136)
    T* partials;       // gridDim.x x lr_len(n)
    size_t m;
    int n;
    int k;
    const T* colscale;      // SCALED kernels: J0 is the finite-difference DIFFERENCE panel D as the caller's kernel wrote it, and
                            // the Jacobian entry is D_ij (1 / twh_j) (0 for a collapsed interval) -- colscale = twh. The same
                            // multiplication k_jtj_fdp performs (LS:1046-1047), applied at load time: the values are bit-identical
                            // to a materialised J, and the refresh does not have to write m x n doubles
    const int32_t* guard;   // optional: the kernel does nothing when *guard == 0 (rounds enqueued ahead of time)
    // ---- fused tail (tail_counters != nullptr): the reduction of k_lr_reduce and, without an all-reduce behind it, the
    //      n x n work of k_lr_finish run inside the sweep, in the workgroups that arrive last (see lr_tail below)
    uint32_t* tail_counters;   // kReduceRanges group counters + 1 top counter: zero before the launch, reset by the tail
    T* range_sums;             // kReduceRanges x lr_len(n)
    T* out;                    // the reduced sweep vector (lr_len(n))
    int finish;                // 1: also apply k_lr_finish (single GPU); 0: an all-reduce of `out` follows
    T* Dw;                     // finish: D (writable: row k receives dx)
    T* JJ; T* Jy; LmState<T>* st;
};


// ---- the tail of the sweep. Workgroup b belongs to range r = b / per (per = ceil(nblk / 32): the ranges of k_lr_reduce).
// Every workgroup publishes its partial vector and counts itself into its range; the LAST arrival of a range sums the
// range's partials in index order into range_sums[r] and counts the range in; the last range to arrive sums the 32 range
// vectors in index order. Same partition and same order as k_lr_reduce whichever workgroups happen to come last: bitwise
// reproducible and bit-identical to the two-kernel path. With `finish` the final workgroup goes on with k_lr_finish's
// work (v = v0 + D w, the rank-two update of J^T J, J^T y, |J^T y|_inf, D_k = dx) from the vector it holds in LDS.
template <typename T>
__device__ inline void lr_tail(const LrArgs<T>& a, T (*red)[lr_len(kLrMaxN)])
{
    const int len = lr_len(a.n), nblk = (int)gridDim.x, tid = threadIdx.x;
    const int per = (nblk + kReduceRanges - 1) / kReduceRanges;
    const int r = (int)blockIdx.x / per;
    const int b0 = r * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
    const int ngroups = (nblk + per - 1) / per;
    if (!arrive_last(&a.tail_counters[r], (uint32_t)(b1 - b0))) return;
    constexpr int kInFlight = 8;                           // loads in flight per thread: the tail must fit the registers of the main loop
#pragma unroll 1
    for (int e = tid; e < len; e += blockDim.x) {
        T p[kInFlight];
        T s = 0;
#pragma unroll 1
        for (int c0 = b0; c0 < b1; c0 += kInFlight) {
            const T* src = a.partials + (size_t)c0 * len + e;
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) p[u] = (c0 + u < b1) ? load_agent(src + (size_t)u * len) : T(0);
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) if (c0 + u < b1) s += p[u];
        }
        store_agent(&a.range_sums[(size_t)r * len + e], s);
    }
    if (!arrive_last(&a.tail_counters[kReduceRanges], (uint32_t)ngroups)) return;
    T* lrs = red[0];                                       // the reduced vector, kept in LDS for the finish
#pragma unroll 1
    for (int e = tid; e < len; e += blockDim.x) {
        T p[kInFlight];
        T tot = 0;
#pragma unroll 1
        for (int c0 = 0; c0 < kReduceRanges; c0 += kInFlight) {
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) p[u] = (c0 + u < ngroups) ? load_agent(&a.range_sums[(size_t)(c0 + u) * len + e]) : T(0);
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) tot = (c0 + u == 0) ? p[0] : tot + p[u];     // k_lr_reduce's order: part[0] + part[1] + ...
        }
        a.out[e] = tot;
        lrs[e] = tot;
    }
    if (!a.finish) return;
    __syncthreads();
    const int n = a.n, k = a.k;
    T* v = red[1];
    T* dxs = red[2];
    const T* w = lrs + 2 * n;
    const T* h = w + kLrMax;
    const T uu = lrs[2 * n + 2 * kLrMax], uy = lrs[2 * n + 2 * kLrMax + 1];
    T mx = 0;
    for (int j = tid; j < n; j += blockDim.x) {
        const T dj = a.dx[j];
        T s = lrs[j];
        for (int l = 0; l < k; ++l) s += a.Dw[(size_t)l * n + j] * w[l];
        v[j] = s;
        dxs[j] = dj;
        T g = lrs[n + j];
        for (int l = 0; l < k; ++l) g += a.Dw[(size_t)l * n + j] * h[l];
        g += dj * uy;
        a.Jy[j] = g;
        const T av = dabs(g);
        if (av > mx) mx = av;
    }
    __syncthreads();                                       // also: every read of D rows < k is done before row k is written
    for (int j = tid; j < n; j += blockDim.x) a.Dw[(size_t)k * n + j] = dxs[j];
    const int nn = n * n;
    for (int base = tid; base < nn; base += 8 * (int)blockDim.x) {         // loads in flight: see lr_finish_block
        T old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int idx = base + u * (int)blockDim.x; old[u] = a.JJ[idx < nn ? idx : nn - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * (int)blockDim.x;
            if (idx < nn) {
                const int i = idx / n, j = idx - i * n;
                const int rr = i >= j ? i : j, cc = i >= j ? j : i;
                a.JJ[idx] = old[u] + lr_jj_term(v[rr], v[cc], dxs[rr], dxs[cc], uu);
            }
        }
    }
    mx = wave_max(mx);
    __shared__ T mred[4];
    if ((tid & 63) == 0) mred[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        T m2 = mred[0];
        for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) m2 = mred[wv] > m2 ? mred[wv] : m2;
        a.st->jy_inf = m2;
    }
}

// (four workgroups per CU up to n = 128: the scaled variant would otherwise take 130 VGPRs and drop to three)
template <typename T, int NCP, bool VEC, bool SCALED = false>
__global__ __launch_bounds__(256, (NCP <= 4 ? 4 : 1)) void k_broyden_lr(const LrArgs<T> a)
{
    using P2 = typename LrPair<T>::type;
    __shared__ T red[4][lr_len(kLrMaxN)];
    __shared__ T coef[kLrMax];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, p = lane & 15;
    const int n = a.n, k = a.k;
    const size_t m = a.m;
    if (a.guard && *a.guard == 0) return;

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
    T sc0[SCALED ? NCP : 1], sc1[SCALED ? NCP : 1];        // 1 / twh of the lane's columns (0: collapsed interval, zero column)
    if constexpr (SCALED) {
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            const T t0 = a.colscale[ok0[c] ? coff[c] : 0], t1 = a.colscale[ok1[c] ? coff[c] + 1 : 0];
            sc0[c] = (t0 == 0 || !ok0[c]) ? T(0) : T(1) / t0;
            sc1[c] = (t1 == 0 || !ok1[c]) ? T(0) : T(1) / t1;
        }
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
    T wl = 0, hl = 0, uu = 0, uy = 0;

    const size_t G = (m + 3) / 4;
    const size_t nw = (size_t)gridDim.x * 4, wid = (size_t)blockIdx.x * 4 + wave;
    const size_t per = (G + nw - 1) / nw;
    const size_t gb = wid * per;
    const size_t ge = gb + per < G ? gb + per : G;
    for (size_t g = gb; g < ge; ++g) {
        const size_t row = 4 * g + q;
        const bool rok = row < m;
        const size_t rr = rok ? row : m - 1;
        const T* __restrict__ rp = a.J + rr * (size_t)n;
        T v0[NCP], v1[NCP];
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            if constexpr (VEC) {
                typedef T lr_v2 __attribute__((ext_vector_type(2)));
                const lr_v2 t = __builtin_nontemporal_load(reinterpret_cast<const lr_v2*>(rp + coff[c]));   // J is swept once per pass and is larger than the Infinity Cache
                v0[c] = t.x;
                v1[c] = t.y;
            } else {
                v0[c] = rp[coff[c]];
                v1[c] = rp[ok1[c] ? coff[c] + 1 : 0];
            }
            if constexpr (SCALED) {
                v0[c] = sc0[c] == 0 ? T(0) : v0[c] * sc0[c];              // scal(1 / twh, Jj), LS:1047 (LS:1046: zero column)
                v1[c] = sc1[c] == 0 ? T(0) : v1[c] * sc1[c];
            }
        }
        T yn = a.y[rr];
        const T yo = a.y_old[rr];
        const T ul = own ? Up[rr] : T(0);
        T s = ul * cp;
#pragma unroll
        for (int c = 0; c < NCP; ++c) s += v0[c] * d0[c] + v1[c] * d1[c];
        s = sum16(s);
        T u = nd * ((yo - yn) + s);                        // LS:1003-1005: axpy(-1, y, mBuffer); gemv; scal(-d)
        if (!rok) { u = 0; yn = 0; }
        if (rok && p == 0) Uk[row] = u;
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            va0[c] += v0[c] * u;
            va1[c] += v1[c] * u;
            ga0[c] += v0[c] * yn;
            ga1[c] += v1[c] * yn;
        }
        wl += ul * u;
        hl += ul * yn;
        uu += u * u;
        uy += u * yn;
    }

    // the four row groups of the wave, then the four waves of the block, in a fixed order
    auto qsum = [](T v) { v += wave_shfl_xor(v, 16); v += wave_shfl_xor(v, 32); return v; };
#pragma unroll
    for (int c = 0; c < NCP; ++c) { va0[c] = qsum(va0[c]); va1[c] = qsum(va1[c]); ga0[c] = qsum(ga0[c]); ga1[c] = qsum(ga1[c]); }
    wl = qsum(wl); hl = qsum(hl); uu = qsum(uu); uy = qsum(uy);
    if (q == 0) {
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            if (ok0[c]) { red[wave][coff[c]] = va0[c]; red[wave][n + coff[c]] = ga0[c]; }
            if (ok1[c]) { red[wave][coff[c] + 1] = va1[c]; red[wave][n + coff[c] + 1] = ga1[c]; }
        }
        red[wave][2 * n + p] = wl;
        red[wave][2 * n + kLrMax + p] = hl;
        if (p == 0) { red[wave][2 * n + 2 * kLrMax] = uu; red[wave][2 * n + 2 * kLrMax + 1] = uy; }
    }
    __syncthreads();
    const int len = lr_len(n);
    T* out = a.partials + (size_t)blockIdx.x * len;
    if (a.tail_counters) {
        // the partial vector crosses to another workgroup inside this kernel: agent-scope stores (common.h)
        for (int e = threadIdx.x; e < len; e += blockDim.x) store_agent(&out[e], (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]));
        lr_tail<T>(a, red);
    } else {
        for (int e = threadIdx.x; e < len; e += blockDim.x) out[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}

// sum the per-block partial vectors in a fixed order: blockDim = 1024 = 32 entries x 32 block ranges (a thread walks
// nparts / 32 partials with 8 loads in flight: the 8-range version spent 9 us on 16 dependent L2 round trips per thread)
template <typename T>
__global__ __launch_bounds__(32 * kReduceRanges) void k_lr_reduce(const T* __restrict__ partials, int nparts, int len, T* __restrict__ out,
                                                                  const int32_t* guard = nullptr)
{
    if (guard && *guard == 0) return;
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

// n x n side of a pass: JJ += v dx^T + dx v^T + uu dx dx^T (block i < n: row i), Jy and |Jy|_inf (LS:1052-1053),
// D_k = dx (block n). `lr` is the (all-reduced) vector of k_lr_reduce.
template <typename T>
__global__ __launch_bounds__(256) void k_lr_finish(const T* __restrict__ lr, T* __restrict__ D, const T* __restrict__ dx,
                                                   int k, int n, T* __restrict__ JJ, T* __restrict__ Jy, LmState<T>* st,
                                                   const int32_t* guard = nullptr)
{
    if (guard && *guard == 0) return;
    __shared__ T v[kLrMaxN];
    __shared__ T red[4];
    const T* __restrict__ w = lr + 2 * n;
    const T* __restrict__ h = w + kLrMax;
    const T uu = lr[2 * n + 2 * kLrMax], uy = lr[2 * n + 2 * kLrMax + 1];
    const int i = blockIdx.x;
    if (i < n) {
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            T s = lr[j];
            for (int l = 0; l < k; ++l) s += D[(size_t)l * n + j] * w[l];
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
        for (int l = 0; l < k; ++l) s += D[(size_t)l * n + j] * h[l];
        s += dx[j] * uy;
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
// src: what J is read from -- J itself (in place), or the unscaled difference panel with colscale = twh (SCALED, see LrArgs)
template <typename T, int NCP, bool VEC, bool SCALED = false>
__global__ __launch_bounds__(256) void k_lr_flush(T* J, const T* __restrict__ U, const T* __restrict__ D,
                                                  int k, size_t m, int n, const T* src = nullptr, const T* colscale = nullptr)
{
    if (!src) src = J;
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
    T sc0[SCALED ? NCP : 1], sc1[SCALED ? NCP : 1];
    if constexpr (SCALED) {
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            const T t0 = colscale[ok0[c] ? coff[c] : 0], t1 = colscale[ok1[c] ? coff[c] + 1 : 0];
            sc0[c] = (t0 == 0 || !ok0[c]) ? T(0) : T(1) / t0;
            sc1[c] = (t1 == 0 || !ok1[c]) ? T(0) : T(1) / t1;
        }
    }
    const size_t G = (m + 3) / 4;
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t g = wave_id; g < G; g += nwaves) {
        const size_t row = 4 * g + q;
        const bool rok = row < m;
        const size_t rr = rok ? row : m - 1;
        T* rp = J + rr * (size_t)n;
        const T* sp = src + rr * (size_t)n;
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
            if constexpr (SCALED) {
                v0[c] = sc0[c] == 0 ? T(0) : v0[c] * sc0[c];
                v1[c] = sc1[c] == 0 ? T(0) : v1[c] * sc1[c];
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

// ---- host-side launchers
template <typename T, int NCP>
hipError_t lr_sweep_ncp(const LrArgs<T>& a, int nblk, bool vec, hipStream_t s)
{
    if (a.colscale) {
        // only the f64, even-n shapes of the difference panel reach here (16-byte loads)
        if constexpr (sizeof(T) == 8) {
            if (!vec) return hipErrorInvalidValue;
            MIRLSQ_LAUNCH((k_broyden_lr<T, NCP, true, true>), dim3(nblk), dim3(256), 0, s, a);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (vec) MIRLSQ_LAUNCH((k_broyden_lr<T, NCP, true>), dim3(nblk), dim3(256), 0, s, a);
    else MIRLSQ_LAUNCH((k_broyden_lr<T, NCP, false>), dim3(nblk), dim3(256), 0, s, a);
    return hipGetLastError();
}
template <typename T>
hipError_t lr_sweep(const LrArgs<T>& a, int nblk, hipStream_t s)
{
    if (a.n > kLrMaxN || a.k < 0 || a.k >= kLrMax) return hipErrorInvalidValue;
    const bool vec = (a.n % 2 == 0) && (reinterpret_cast<uintptr_t>(a.J) % (2 * sizeof(T)) == 0);
    const int ncp = (a.n + 31) / 32;
    if (ncp <= 1) return lr_sweep_ncp<T, 1>(a, nblk, vec, s);
    if (ncp <= 2) return lr_sweep_ncp<T, 2>(a, nblk, vec, s);
    if (ncp <= 4) return lr_sweep_ncp<T, 4>(a, nblk, vec, s);
    return lr_sweep_ncp<T, 8>(a, nblk, vec, s);
}

template <typename T, int NCP>
hipError_t lr_flush_ncp(T* J, const T* U, const T* D, int k, size_t m, int n, int nblk, bool vec, hipStream_t s, const T* src, const T* colscale)
{
    const size_t lds = (size_t)k * n * sizeof(T);
    if (colscale) {
        if constexpr (sizeof(T) == 8) {
            if (!vec) return hipErrorInvalidValue;
            MIRLSQ_LAUNCH((k_lr_flush<T, NCP, true, true>), dim3(nblk), dim3(256), lds, s, J, U, D, k, m, n, src, colscale);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (vec) MIRLSQ_LAUNCH((k_lr_flush<T, NCP, true>), dim3(nblk), dim3(256), lds, s, J, U, D, k, m, n, src, colscale);
    else MIRLSQ_LAUNCH((k_lr_flush<T, NCP, false>), dim3(nblk), dim3(256), lds, s, J, U, D, k, m, n, src, colscale);
    return hipGetLastError();
}
// J <- src (scaled by 1 / colscale when given) + the k pending terms; src == nullptr: J in place. With k == 0 and a scaled
// source this just materialises J.
template <typename T>
hipError_t lr_flush(T* J, const T* U, const T* D, int k, size_t m, int n, int num_cu, hipStream_t s, const T* src = nullptr,
                    const T* colscale = nullptr)
{
    if (k <= 0 && !colscale) return hipSuccess;
    if (n > kLrMaxN || k > kLrMax) return hipErrorInvalidValue;
    const size_t G = (m + 3) / 4;
    size_t blocks = (G + 3) / 4;
    if (blocks > (size_t)num_cu * 8) blocks = (size_t)num_cu * 8;
    if (blocks < 1) blocks = 1;
    const bool vec = (n % 2 == 0) && (reinterpret_cast<uintptr_t>(J) % (2 * sizeof(T)) == 0)
        && (reinterpret_cast<uintptr_t>(src) % (2 * sizeof(T)) == 0);
    const int ncp = (n + 31) / 32;
    if (ncp <= 1) return lr_flush_ncp<T, 1>(J, U, D, k, m, n, (int)blocks, vec, s, src, colscale);
    if (ncp <= 2) return lr_flush_ncp<T, 2>(J, U, D, k, m, n, (int)blocks, vec, s, src, colscale);
    if (ncp <= 4) return lr_flush_ncp<T, 4>(J, U, D, k, m, n, (int)blocks, vec, s, src, colscale);
    return lr_flush_ncp<T, 8>(J, U, D, k, m, n, (int)blocks, vec, s, src, colscale);
}

// workgroups of the sweep: 4 per CU, at least ~8 row steps per wave
inline int lr_blocks(size_t m, int num_cu)
{
    const size_t G = (m + 3) / 4;
    size_t want = (G + 4 * 8 - 1) / (4 * 8);
    const size_t cap = (size_t)num_cu * 4;
    if (want > cap) want = cap;
    return (int)(want ? want : 1);
}

}  // namespace mirlsq
