// jtj_wide.h -- J^T J (lower) + J^T y (+ Broyden) for 128 < n <= 256 (BASELINE cfg 4: n = 256).
//
// Same fragment scheme as jtj_kernel.h (lane (q, p) holds J[4g + q][16 c + p] = MFMA A and B operand), but the
// 16 x 16-block output grid no longer fits one workgroup's registers, so it is cut into 4 x 4-block tiles
// (64 x 64 outputs, 16 accumulators per wave): job (TI, TJ), TJ <= TI, streams all rows and loads only the
// column blocks of its two panels. Diagonal jobs also produce their panel's part of J^T y.
// The Broyden rank-1 update (LS:1003-1006) needs the whole row for its dot product, so it is a separate
// HBM-bound pass (k_broyden_wide) in front of the products.
// Register streaming like k_jtj, with a counted ring of row groups in flight (round 4: 0.56 of the MFMA peak at n = 512).
#pragma once

#include "common.h"
#include "jtj_plan.h"

namespace mirlsq {


template <typename T>
struct JtjWideArgs {
    const T* J;
    const T* y;
    T* slabs;          // [job][workgroup][kWideSlabLen]
    size_t m;
    int n;
    int nt;            // tiles per side = ceil(ceil(n / 16) / 4)
};

__host__ __device__ inline void wide_job_to_tile(int job, int& ti, int& tj)
{
    ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= job) ++ti;
    tj = job - ti * (ti + 1) / 2;
}

template <typename T>
__global__ __launch_bounds__(256) void k_jtj_wide(JtjWideArgs<T> a)
{
    using Acc = typename Mma<T>::Acc;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    T* red = reinterpret_cast<T*>(smem_w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, p = lane & 15;
    int ti, tj;
    wide_job_to_tile(blockIdx.y, ti, tj);
    const bool diag = ti == tj;
    const size_t m = a.m;
    const int n = a.n;

    const size_t G = (m + 3) / 4;
    const size_t nslots = (size_t)gridDim.x * 4;
    const size_t per = (G + nslots - 1) / nslots;
    const size_t slot = (size_t)blockIdx.x * 4 + wave;
    const size_t g0 = slot * per < G ? slot * per : G;
    const size_t g1 = g0 + per < G ? g0 + per : G;

    Acc acc[kWideTile][kWideTile];
#pragma unroll
    for (int i = 0; i < kWideTile; ++i)
#pragma unroll
        for (int j = 0; j < kWideTile; ++j) acc[i][j] = Acc{0, 0, 0, 0};
    T jy[kWideTile];
    int ci[kWideTile], cj[kWideTile];
    bool oki[kWideTile], okj[kWideTile];
#pragma unroll
    for (int c = 0; c < kWideTile; ++c) {
        jy[c] = 0;
        const int coli = 16 * (kWideTile * ti + c) + p, colj = 16 * (kWideTile * tj + c) + p;
        oki[c] = coli < n; okj[c] = colj < n;
        ci[c] = oki[c] ? coli : n - 1;
        cj[c] = okj[c] ? colj : n - 1;
    }
    // a fragment holds what was LOADED; rows past m and columns past n are zeroed when it is used (a select right behind the load
    // would wait for the load where it is issued and serialise the ring)
    struct Frag { T vi[kWideTile], vj[kWideTile], y; bool rok; };
    auto load = [&](size_t g, Frag& f) {
        const size_t row = 4 * g + q;
        f.rok = row < m;
        const size_t rc = f.rok ? row : m - 1;
        const T* rp = a.J + rc * (size_t)n;
#pragma unroll
        for (int c = 0; c < kWideTile; ++c) f.vi[c] = rp[ci[c]];
#pragma unroll
        for (int c = 0; c < kWideTile; ++c) f.vj[c] = rp[cj[c]];
        f.y = a.y[rc];
    };
    auto compute = [&](const Frag& f) {
        T vi[kWideTile], vj[kWideTile];
#pragma unroll
        for (int c = 0; c < kWideTile; ++c) { vi[c] = (f.rok && oki[c]) ? f.vi[c] : T(0); vj[c] = (f.rok && okj[c]) ? f.vj[c] : T(0); }
        const T yv = f.rok ? f.y : T(0);
        if (diag) {
#pragma unroll
            for (int c = 0; c < kWideTile; ++c) jy[c] += vi[c] * yv;       // LS:1052
        }
#pragma unroll
        for (int i = 0; i < kWideTile; ++i)
#pragma unroll
            for (int j = 0; j < kWideTile; ++j) acc[i][j] = Mma<T>::mma(vi[i], vj[j], acc[i][j]);   // LS:1065
    };
    // Ring of kRing fragments, kAhead row groups in flight beyond the one on the matrix cores. The loads of a group are issued
    // UNCONDITIONALLY (past the end the last group is read again): a load under `if` makes the number of loads in flight
    // unknowable to the compiler, which then waits for all of them before the first use -- one memory latency per row group with
    // two waves a SIMD was 0.16 of the MFMA peak at n = 512 (the rule found on the n = 256 solve).
    constexpr int kAhead = 3, kRing = kAhead + 1;
    Frag fr_[kRing];
    if (g0 < g1) {
        auto issue = [&](size_t g, auto B) { load(g < g1 ? g : g1 - 1, fr_[decltype(B)::value]); };
        static_for<kAhead>([&](auto U) { issue(g0 + decltype(U)::value, U); });
        for (size_t gb = g0; gb < g1; gb += kRing) {
            static_for<kRing>([&](auto U) {
                constexpr int u = decltype(U)::value;
                if (gb + u < g1) {
                    issue(gb + u + kAhead, IntC<(u + kAhead) % kRing>{});
                    compute(fr_[u]);
                }
            });
        }
    }
#pragma unroll
    for (int c = 0; c < kWideTile; ++c) {
        jy[c] += wave_shfl_xor(jy[c], 16);
        jy[c] += wave_shfl_xor(jy[c], 32);
    }
    // workgroup reduction through LDS in a fixed order: (w0 + w2) + (w1 + w3)
    constexpr int SL = kWideSlabLen;
    auto put = [&](T* dst) {
#pragma unroll
        for (int i = 0; i < kWideTile; ++i)
#pragma unroll
            for (int j = 0; j < kWideTile; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((i * kWideTile + j) * 4 + r) * kWave + lane] = acc[i][j][r];
#pragma unroll
        for (int c = 0; c < kWideTile; ++c) dst[(kWideTile * kWideTile * 4 + c) * kWave + lane] = jy[c];
    };
    auto add = [&](const T* src) {
#pragma unroll
        for (int i = 0; i < kWideTile; ++i)
#pragma unroll
            for (int j = 0; j < kWideTile; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += src[((i * kWideTile + j) * 4 + r) * kWave + lane];
#pragma unroll
        for (int c = 0; c < kWideTile; ++c) jy[c] += src[(kWideTile * kWideTile * 4 + c) * kWave + lane];
    };
    if (wave >= 2) put(red + (size_t)(wave - 2) * SL);
    __syncthreads();
    if (wave < 2) add(red + (size_t)wave * SL);
    __syncthreads();
    if (wave == 1) put(red);
    __syncthreads();
    if (wave == 0) {
        add(red);
        put(a.slabs + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SL);
    }
}

// fixed-order sum over workgroups, scatter into packed [JJ lower | Jy]. grid = (ceil(SL / 32), jobs), 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void k_jtj_wide_reduce(const T* __restrict__ slabs, int nwg, int n, T* __restrict__ packed)
{
    __shared__ T part[8][32];
    constexpr int SL = kWideSlabLen;
    const int es = threadIdx.x & 31, sp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + es;
    const int job = blockIdx.y;
    T s = 0;
    if (e < SL) {
        const int per = (nwg + 7) / 8;
        const int b0 = sp * per, b1 = (b0 + per < nwg) ? b0 + per : nwg;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += slabs[((size_t)job * nwg + b) * SL + e];
    }
    part[sp][es] = s;
    __syncthreads();
    if (sp == 0 && e < SL) {
        T tot = part[0][es];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += part[k][es];
        int ti, tj;
        wide_job_to_tile(job, ti, tj);
        const int reg = e / kWave, lane = e % kWave;
        if (reg < kWideTile * kWideTile * 4) {
            const int blk = reg >> 2, r = reg & 3;
            const int bi = blk / kWideTile, bj = blk % kWideTile;
            const int row = 16 * (kWideTile * ti + bi) + Mma<T>::row(lane, r);
            const int col = 16 * (kWideTile * tj + bj) + (lane & 15);
            if (row < n && col <= row) packed[(size_t)row * (row + 1) / 2 + col] = tot;
        } else if (ti == tj) {
            const int c = reg - kWideTile * kWideTile * 4;
            const int col = 16 * (kWideTile * ti + c) + lane;
            if (lane < 16 && col < n) packed[(size_t)n * (n + 1) / 2 + col] = tot;
        }
    }
}

// Broyden update for wide rows, LS:1002-1006: J[i,:] += ((y_i - yold_i - J[i,:].dx) / dx.dx) dx^T. HBM-bound.
template <typename T>
__global__ __launch_bounds__(256) void k_broyden_wide(T* __restrict__ J, const T* __restrict__ y, const T* __restrict__ y_old,
                                                      const T* __restrict__ dx, const T* __restrict__ dx_dot, size_t m, int n)
{
    constexpr int NCBW = 16;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    T dxr[NCBW];
    int co[NCBW];
    bool ok[NCBW];
#pragma unroll
    for (int c = 0; c < NCBW; ++c) {
        const int col = 16 * c + p;
        ok[c] = col < n;
        co[c] = ok[c] ? col : n - 1;
        const T t = dx[co[c]];
        dxr[c] = ok[c] ? t : T(0);
    }
    const T neg_d = -(T(1) / *dx_dot);
    const size_t G = (m + 3) / 4;
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t g = wave_id; g < G; g += nwaves) {
        const size_t row = 4 * g + q;
        const bool rok = row < m;
        T* rp = J + (rok ? row : m - 1) * (size_t)n;
        T v[NCBW];
#pragma unroll
        for (int c = 0; c < NCBW; ++c) { const T t = rp[co[c]]; v[c] = (rok && ok[c]) ? t : T(0); }
        const T yv = rok ? y[rok ? row : 0] : T(0), yo = rok ? y_old[rok ? row : 0] : T(0);
        T part = 0;
#pragma unroll
        for (int c = 0; c < NCBW; ++c) part += v[c] * dxr[c];
        part = sum16(part);
        const T t = (yo - yv) + part;
        const T u = neg_d * t;
#pragma unroll
        for (int c = 0; c < NCBW; ++c) if (rok && ok[c]) rp[co[c]] = v[c] + u * dxr[c];
    }
}

// The same update for ANY n (used above n = 256): one wave per row, lanes along the row (coalesced), two sweeps over the
// row (the dot product, then the update; the second one hits L1 / L2).
template <typename T>
__global__ __launch_bounds__(256) void k_broyden_rows(T* __restrict__ J, const T* __restrict__ y, const T* __restrict__ y_old,
                                                      const T* __restrict__ dx, const T* __restrict__ dx_dot, size_t m, int n)
{
    const int lane = threadIdx.x & 63;
    const T neg_d = -(T(1) / *dx_dot);
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t row = wave_id; row < m; row += nwaves) {
        T* rp = J + row * (size_t)n;
        T part = 0;
        for (int c = lane; c < n; c += kWave) part += rp[c] * dx[c];
        part = wave_sum(part);
        const T t = (y_old[row] - y[row]) + part;      // LS:1003-1004
        const T u = neg_d * t;                         // LS:1005
        for (int c = lane; c < n; c += kWave) rp[c] = rp[c] + u * dx[c];   // LS:1006
    }
}

}  // namespace mirlsq
