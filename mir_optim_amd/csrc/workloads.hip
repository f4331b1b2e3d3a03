// workloads.hip -- DEVICE residual callbacks for the synthetic workloads of SURVEY.md section 8d
// (the "user code" side of the boundary: these play the role of the caller's f / g / batched f,
// LeastSquaresFunctionBetterC at /root/reference/source/mir/optim/least_squares.d:78-80, with the
// device-pointer contract of MIR_LSQ_DEVICE_CALLBACKS). Built as a separate shared library
// (libmir_optim_amd_workloads.so) so the solver library stays free of model code.
//
// Also holds the host-side counter RNG u(k) = (splitmix64(seed + k) >> 11) * 2^-53 used to build
// bit-identical inputs for the GPU run and the CPU baseline.
#include "workloads_device.h"
#include "workloads_gemm.h"

namespace {

// ---- tanh-linear: y_i = tanh(a_i . x) - b_i.  A wave handles 4 rows per step; lane (q, p) reads the PAIRS
//      A[4g + q][32 c + 2p, + 1] with one 16-byte load (8 bytes for f32): every 16-lane group streams 256
//      contiguous bytes per load. n even: vector loads; n odd: the scalar layout (pair split into two loads).
//      MODE 0: residual; MODE 1: analytic Jacobian row (1 - tanh^2) a_i.
template <typename T> struct Pair;
template <> struct Pair<double> { using type = double2; };
template <> struct Pair<float> { using type = float2; };

template <typename T, int NCP, int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_tanh_linear(const T* __restrict__ A, const T* __restrict__ b,
                                                     const T* __restrict__ x, T* __restrict__ out, size_t m, int n)
{
    using P2 = typename Pair<T>::type;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    T x0[NCP], x1[NCP];
    int coff[NCP];
    bool ok0[NCP], ok1[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
        const int col = 32 * c + 2 * p;
        ok0[c] = col < n;
        ok1[c] = col + 1 < n;
        coff[c] = ok0[c] ? col : 0;
        const T t0 = x[ok0[c] ? col : 0], t1 = x[ok1[c] ? col + 1 : 0];
        x0[c] = ok0[c] ? t0 : T(0);
        x1[c] = ok1[c] ? t1 : T(0);
    }
    const size_t G = (m + 3) / 4;
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    // a contiguous range of row groups per wave (like the library's sweep over J: 6.2-6.4 TB/s against 5.9 for the
    // grid-stride walk this kernel used to do)
    const size_t per = (G + nwaves - 1) / nwaves;
    const size_t gb = wave_id * per < G ? wave_id * per : G;
    const size_t ge = gb + per < G ? gb + per : G;
    // A ring of row groups in flight: the loads of groups g + 1 .. g + kAhead are issued (unconditionally, indices clamped: counted
    // waits) before group g is summed. With one group in flight a wave had 4 KB on its way and the sweep ran at 6.0-6.3 TB/s; a plain
    // streaming read of the same gigabyte reaches 7.1 on this part (scripts/probes/copy_bw_probe.hip).
    constexpr int kAhead = (VEC && MODE == 0) ? 3 : 0, kRing = kAhead + 1;
    T v0[kRing][NCP], v1[kRing][NCP], bv[kRing];
    auto load = [&](size_t g, auto B) {
        constexpr int buf = decltype(B)::value;
        const size_t gc = g < ge ? g : (ge > gb ? ge - 1 : gb);
        const size_t row = 4 * gc + q;
        const T* rp = A + (row < m ? row : m - 1) * (size_t)n;
        if constexpr (MODE == 0) bv[buf] = b[row < m ? row : m - 1];   // with the group's loads, not under the store's `if`: a load
                                                                       // there drains the ring (s_waitcnt vmcnt(0) per group)
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            if constexpr (VEC) {                           // n even: a pair never straddles the row end
                typedef T wl_v2 __attribute__((ext_vector_type(2)));
                const wl_v2 t = __builtin_nontemporal_load(reinterpret_cast<const wl_v2*>(rp + coff[c]));   // A is swept once per call
                v0[buf][c] = t.x;
                v1[buf][c] = t.y;
            } else {
                v0[buf][c] = rp[coff[c]];
                v1[buf][c] = rp[ok1[c] ? coff[c] + 1 : 0];
            }
        }
    };
    auto use = [&](size_t g, auto B) {
        constexpr int buf = decltype(B)::value;
        const size_t row = 4 * g + q;
        const bool rok = row < m;
        T s = 0;
#pragma unroll
        for (int c = 0; c < NCP; ++c) s += v0[buf][c] * x0[c] + v1[buf][c] * x1[c];
        s = sum16(s);
        const T t = dtanh(s);
        if constexpr (MODE == 0) {
            if (rok && p == 0) out[row] = t - bv[buf];
        } else {
            const T d = 1 - t * t;
            T* op = out + (rok ? row : m - 1) * (size_t)n;
#pragma unroll
            for (int c = 0; c < NCP; ++c) {
                if (rok && ok0[c]) op[coff[c]] = d * v0[buf][c];
                if (rok && ok1[c]) op[coff[c] + 1] = d * v1[buf][c];
            }
        }
    };
    if (gb >= ge) return;
    wl_static_for<kAhead>([&](auto U) { load(gb + decltype(U)::value, U); });
    for (size_t g0 = gb; g0 < ge; g0 += kRing) {
        wl_static_for<kRing>([&](auto U) {
            constexpr int u0 = decltype(U)::value;
            load(g0 + u0 + kAhead, std::integral_constant<int, (u0 + kAhead) % kRing>{});
            if (g0 + u0 < ge) use(g0 + u0, U);
        });
    }
}

// any n (used above n = 256): one wave per row, lanes along the row
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_tanh_linear_rows(const T* __restrict__ A, const T* __restrict__ b,
                                                          const T* __restrict__ x, T* __restrict__ out, size_t m, int n)
{
    const int lane = threadIdx.x & 63;
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t row = wave_id; row < m; row += nwaves) {
        const T* rp = A + row * (size_t)n;
        T s = 0;
        for (int c = lane; c < n; c += 64) s += rp[c] * x[c];
        s = sum16(s);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const T t = dtanh(s);
        if constexpr (MODE == 0) {
            if (lane == 0) out[row] = t - b[row];
        } else {
            const T d = 1 - t * t;
            T* op = out + row * (size_t)n;
            for (int c = lane; c < n; c += 64) op[c] = d * rp[c];
        }
    }
}

template <typename T, int NCP, int MODE>
void launch_tanh_linear_ncp(const T* A, const T* b, const T* x, T* out, size_t m, int n, dim3 grid, hipStream_t s)
{
    const bool vec = (n % 2 == 0) && (reinterpret_cast<uintptr_t>(A) % (2 * sizeof(T)) == 0);
    if (vec) hipLaunchKernelGGL((k_tanh_linear<T, NCP, MODE, true>), grid, dim3(256), 0, s, A, b, x, out, m, n);
    else hipLaunchKernelGGL((k_tanh_linear<T, NCP, MODE, false>), grid, dim3(256), 0, s, A, b, x, out, m, n);
}

template <typename T, int MODE>
void launch_tanh_linear(const T* A, const T* b, const T* x, T* out, size_t m, int n, hipStream_t s)
{
    const size_t G = (m + 3) / 4;
    size_t blocks = (G + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;
    if (blocks < 1) blocks = 1;
    const int ncp = (n + 31) / 32;                         // column pairs per lane
    dim3 grid((unsigned)blocks);
    if (ncp > 8) {
        hipLaunchKernelGGL((k_tanh_linear_rows<T, MODE>), grid, dim3(256), 0, s, A, b, x, out, m, n);
        return;
    }
    if (ncp <= 1) launch_tanh_linear_ncp<T, 1, MODE>(A, b, x, out, m, n, grid, s);
    else if (ncp <= 2) launch_tanh_linear_ncp<T, 2, MODE>(A, b, x, out, m, n, grid, s);
    else if (ncp <= 4) launch_tanh_linear_ncp<T, 4, MODE>(A, b, x, out, m, n, grid, s);
    else launch_tanh_linear_ncp<T, 8, MODE>(A, b, x, out, m, n, grid, s);
}

// ---- Gaussian-sum: n = 3K+1, x = [a | c | w | b]; y_i = sum_k a_k exp(-(t_i-c_k)^2/(2 w_k^2)) + b - data_i
// ONE expression for the single-point and the batched kernels (they must round alike): per Gaussian the constants (a, c, g) with
// g = -1 / (2 w w) -- one division per POINT and Gaussian, not one per row, point and Gaussian: 16 M of them at cfg 2's refresh --
// and per row s = fma(a, exp((d d) g), s), k ascending from s = b.
constexpr int kGaussMax = 16;                        // Gaussians whose constants a thread keeps in registers (cfg 2: 5)
template <typename T> __device__ inline T gauss_g(T w) { return T(-1) / (2 * w * w); }
template <typename T, int KC>
__device__ inline T gauss_row(T ti, const T (&a)[KC], const T (&c)[KC], const T (&g)[KC], int K, T base)
{
    T s = base;
#pragma unroll
    for (int k = 0; k < KC; ++k)
        if (k < K) { const T d = ti - c[k]; s = fma(a[k], dexp((d * d) * g[k]), s); }
    return s;
}
template <typename T> __device__ inline T gauss_row_any(T ti, const T* x, int K)    // more than kGaussMax Gaussians: constants from memory
{
    T s = x[3 * K];
    for (int k = 0; k < K; ++k) { const T d = ti - x[K + k]; s = fma(x[k], dexp((d * d) * gauss_g(x[2 * K + k])), s); }
    return s;
}

template <typename T>
__global__ __launch_bounds__(256) void k_gauss_sum(const T* __restrict__ t, const T* __restrict__ data,
                                                   const T* __restrict__ x, T* __restrict__ y, size_t m, int n)
{
    const int K = (n - 1) / 3;
    if (K <= kGaussMax) {
        T a[kGaussMax], c[kGaussMax], g[kGaussMax];
#pragma unroll
        for (int k = 0; k < kGaussMax; ++k) { a[k] = k < K ? x[k] : T(0); c[k] = k < K ? x[K + k] : T(0); g[k] = k < K ? gauss_g(x[2 * K + k]) : T(0); }
        const T base = x[3 * K];
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x)
            y[i] = gauss_row<T, kGaussMax>(t[i], a, c, g, K, base) - data[i];
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x)
            y[i] = gauss_row_any(t[i], x, K) - data[i];
    }
}

// P points in one launch (the batched callback of mir_lsq_gpu_options): element e <-> (row i, point k);
// RM = false: Y[k m + i] (point-major), RM = true: Y[i P + k] (row-major, the layout of the fused finite-difference kernel).
// FIXED (the launcher's choice when the grid's thread count is a multiple of P, row-major): a thread keeps ONE point for all its
// rows -- its constants are formed once, and the row index advances by a constant instead of a 64-bit division per element.
template <typename T, bool RM, bool FIXED>
__global__ __launch_bounds__(256) void k_gauss_sum_batched(const T* __restrict__ t, const T* __restrict__ data,
                                                           const T* __restrict__ X, T* __restrict__ Y, size_t m, int n, int P)
{
    const int K = (n - 1) / 3;
    const size_t total = m * (size_t)P;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if constexpr (FIXED) {
        static_assert(RM, "a fixed point per thread is the row-major walk");
        const int k = (int)(gid % (size_t)P);
        const T* x = X + (size_t)k * n;
        T a[kGaussMax], c[kGaussMax], g[kGaussMax];
#pragma unroll
        for (int q = 0; q < kGaussMax; ++q) { a[q] = q < K ? x[q] : T(0); c[q] = q < K ? x[K + q] : T(0); g[q] = q < K ? gauss_g(x[2 * K + q]) : T(0); }
        const T base = x[3 * K];
        const size_t rstep = stride / (size_t)P;
        size_t i = gid / (size_t)P;
        for (size_t e = gid; e < total; e += stride, i += rstep) Y[e] = gauss_row<T, kGaussMax>(t[i], a, c, g, K, base) - data[i];
    } else {
        for (size_t e = gid; e < total; e += stride) {
            const size_t i = RM ? e / P : e % m;
            const int k = (int)(RM ? e % P : e / m);
            Y[e] = gauss_row_any(t[i], X + (size_t)k * n, K) - data[i];
        }
    }
}

// ---- exponential decay: kind 0: p0 exp(-t p1) - data ; kind 1: p0 exp(-t / p1) + p2 - data
template <typename T>
__global__ __launch_bounds__(256) void k_exp_decay(const T* __restrict__ t, const T* __restrict__ data,
                                                   const T* __restrict__ x, T* __restrict__ y, size_t m, int kind)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        if (kind == 0) y[i] = x[0] * dexp(-t[i] * x[1]) - data[i];
        else y[i] = x[0] * dexp(-t[i] / x[1]) + x[2] - data[i];
    }
}



// ---- a handful of points (p <= 8) in one sweep over A: the lambda-ladder trials of the solver. HBM-bound like the
//      single-point kernel (A is read once), one 16-lane DPP reduction and one tanh per (row, point).
template <int NCP, int NP>
__global__ __launch_bounds__(256) void k_tanh_linear_multi(const double* __restrict__ A, const double* __restrict__ b,
                                                           const double* __restrict__ X, double* __restrict__ Y,
                                                           size_t m, int n, int P)
{
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, p = lane & 15;
    double x0[NP][NCP], x1[NP][NCP];                       // column pairs 32 c + 2 p, + 1 (n even)
    int coff[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
        const int col = 32 * c + 2 * p;
        const bool ok = col < n;
        coff[c] = ok ? col : 0;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const double2 t = *reinterpret_cast<const double2*>(X + (size_t)(k < P ? k : 0) * n + coff[c]);
            x0[k][c] = (ok && k < P) ? t.x : 0.0;
            x1[k][c] = (ok && k < P) ? t.y : 0.0;
        }
    }
    const size_t G = (m + 3) / 4;
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t g = wave_id; g < G; g += nwaves) {
        const size_t row = 4 * g + q;
        const bool rok = row < m;
        const double* rp = A + (rok ? row : m - 1) * (size_t)n;
        double2 v[NCP];
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            typedef double wl_d2 __attribute__((ext_vector_type(2)));
            const wl_d2 t = __builtin_nontemporal_load(reinterpret_cast<const wl_d2*>(rp + coff[c]));
            v[c] = make_double2(t.x, t.y);
        }
        const double bv = b[rok ? row : m - 1];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (k < P) {
                double s = 0;
#pragma unroll
                for (int c = 0; c < NCP; ++c) s += v[c].x * x0[k][c] + v[c].y * x1[k][c];
                s = sum16(s);
                if (rok && p == 0) Y[(size_t)k * m + row] = dtanh(s) - bv;
            }
        }
    }
}

bool launch_tanh_linear_multi(const double* A, const double* b, const double* X, double* Y, size_t m, int n, int P, hipStream_t s)
{
    if (P > 8 || n > 128 || n % 2 != 0) return false;
    if (reinterpret_cast<uintptr_t>(A) % 16 != 0 || reinterpret_cast<uintptr_t>(X) % 16 != 0) return false;
    const size_t G = (m + 3) / 4;
    size_t blocks = (G + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const int ncp = (n + 31) / 32;
    dim3 grid((unsigned)blocks), blk(256);
    if (ncp <= 1) hipLaunchKernelGGL((k_tanh_linear_multi<1, 8>), grid, blk, 0, s, A, b, X, Y, m, n, P);
    else if (ncp <= 2) hipLaunchKernelGGL((k_tanh_linear_multi<2, 8>), grid, blk, 0, s, A, b, X, Y, m, n, P);
    else hipLaunchKernelGGL((k_tanh_linear_multi<4, 8>), grid, blk, 0, s, A, b, X, Y, m, n, P);
    return true;
}

inline unsigned blocks_for(size_t m)
{
    size_t b = (m + 255) / 256;
    if (b > 256 * 8) b = 256 * 8;
    return (unsigned)(b ? b : 1);
}

inline uint64_t splitmix64_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

__global__ __launch_bounds__(256) void k_busy(long long ticks)
{
    extern __shared__ unsigned char busy_smem[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) busy_smem[0] = 1;                  // (the LDS is really allocated)
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" {

// contexts: device data pointers + the stream the solver was given (mir_lsq_gpu_options.stream)
struct wl_tanh_linear_ctx { const void* A; const void* b; void* stream; int read_a_once; /* difference-panel GEMM: one sweep over A */ };
struct wl_curve_ctx { const void* t; const void* data; void* stream; int kind; };

void wl_tanh_linear_f_d(void* vctx, size_t m, size_t n, const double* x, double* y)
{
    auto* c = static_cast<wl_tanh_linear_ctx*>(vctx);
    launch_tanh_linear<double, 0>((const double*)c->A, (const double*)c->b, x, y, m, (int)n, (hipStream_t)c->stream);
}
void wl_tanh_linear_g_d(void* vctx, size_t m, size_t n, const double* x, double* J)
{
    auto* c = static_cast<wl_tanh_linear_ctx*>(vctx);
    launch_tanh_linear<double, 1>((const double*)c->A, (const double*)c->b, x, J, m, (int)n, (hipStream_t)c->stream);
}
// batched residual (mir_lsq_batched_function_d): p points in one sweep over A (MFMA GEMM); shapes the
// GEMM kernel does not cover fall back to one sweep per point.
void wl_tanh_linear_fb_d(void* vctx, size_t m, size_t n, size_t p, const double* X, double* Y)
{
    auto* c = static_cast<wl_tanh_linear_ctx*>(vctx);
    if (p <= 8 && launch_tanh_linear_multi((const double*)c->A, (const double*)c->b, X, Y, m, (int)n, (int)p, (hipStream_t)c->stream)) return;
    if (launch_tanh_linear_batched((const double*)c->A, (const double*)c->b, X, Y, m, (int)n, (int)p, (hipStream_t)c->stream)) return;
    for (size_t k = 0; k < p; ++k)
        launch_tanh_linear<double, 0>((const double*)c->A, (const double*)c->b, X + k * n, Y + k * m, m, (int)n, (hipStream_t)c->stream);
}
// the same p points with Y written m x p row-major (mir_lsq_gpu_options.fbRowMajor)
void wl_tanh_linear_fbr_d(void* vctx, size_t m, size_t n, size_t p, const double* X, double* Y)
{
    auto* c = static_cast<wl_tanh_linear_ctx*>(vctx);
    launch_tanh_linear_batched_rm((const double*)c->A, (const double*)c->b, X, Y, m, (int)n, (int)p, (hipStream_t)c->stream);
}
// the same p = 2n points [x + h e_0, x - h e_0, ...] with the m x n row-major DIFFERENCE panel written
// (mir_lsq_gpu_options.fbRowMajorDiff): D[i n + j] = f(X_2j)_i - f(X_2j+1)_i
void wl_tanh_linear_fbd_d(void* vctx, size_t m, size_t n, size_t p, const double* X, double* D)
{
    auto* c = static_cast<wl_tanh_linear_ctx*>(vctx);
    launch_tanh_linear_batched_diff((const double*)c->A, (const double*)c->b, X, D, m, (int)n, (int)p, (hipStream_t)c->stream, c->read_a_once);
}
void wl_tanh_linear_f_s(void* vctx, size_t m, size_t n, const float* x, float* y)
{
    auto* c = static_cast<wl_tanh_linear_ctx*>(vctx);
    launch_tanh_linear<float, 0>((const float*)c->A, (const float*)c->b, x, y, m, (int)n, (hipStream_t)c->stream);
}
void wl_tanh_linear_g_s(void* vctx, size_t m, size_t n, const float* x, float* J)
{
    auto* c = static_cast<wl_tanh_linear_ctx*>(vctx);
    launch_tanh_linear<float, 1>((const float*)c->A, (const float*)c->b, x, J, m, (int)n, (hipStream_t)c->stream);
}

// HOST version of the tanh-linear residual (reference contract: x, y are host pointers, LS:78): what a caller of
// the unmodified `mir_optimize_least_squares_d` would pass. ctx: host copies of A and b.
struct wl_tanh_linear_host_ctx { const double* A; const double* b; };
void wl_tanh_linear_f_host_d(void* vctx, size_t m, size_t n, const double* x, double* y)
{
    auto* c = static_cast<wl_tanh_linear_host_ctx*>(vctx);
#pragma omp parallel for schedule(static)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)m; ++i) {
        const double* a = c->A + (size_t)i * n;
        double s = 0;
        for (size_t j = 0; j < n; ++j) s += a[j] * x[j];
        y[i] = std::tanh(s) - c->b[i];
    }
}

// The same residual on ONE host thread: what the tasks of a thread manager call (the manager supplies the parallelism,
// LS:184-215: one finite-difference column per task).
void wl_tanh_linear_f_host_serial_d(void* vctx, size_t m, size_t n, const double* x, double* y)
{
    auto* c = static_cast<wl_tanh_linear_host_ctx*>(vctx);
    for (size_t i = 0; i < m; ++i) {
        const double* a = c->A + i * n;
        double s = 0;
        for (size_t j = 0; j < n; ++j) s += a[j] * x[j];
        y[i] = std::tanh(s) - c->b[i];
    }
}

// A native thread manager with the reference's contract (mir_least_squares_thread_manager, LS:672-678; what the D tier
// builds from a TaskPool, LS:184-215): task(taskContext, totalThreads, threadId, i) for every i in [0, count), threadId <
// totalThreads, a thread runs one task at a time. ctx: optional int* with the number of threads (0 / null: OpenMP's default).
struct wl_task { void* context; void* fn; };          // the 16-byte D delegate, passed by value
typedef void (*wl_task_fn)(wl_task, uint32_t, uint32_t, uint32_t);
void wl_omp_thread_manager(void* ctx, uint32_t count, wl_task task, wl_task_fn fn)
{
    const int want = ctx ? *static_cast<const int*>(ctx) : 0;
#pragma omp parallel num_threads(want > 0 ? want : omp_get_max_threads())
    {
        const uint32_t total = (uint32_t)omp_get_num_threads(), tid = (uint32_t)omp_get_thread_num();
#pragma omp for schedule(dynamic, 1)
        for (ptrdiff_t i = 0; i < (ptrdiff_t)count; ++i) fn(task, total, tid, (uint32_t)i);
    }
}

void wl_gauss_sum_f_d(void* vctx, size_t m, size_t n, const double* x, double* y)
{
    auto* c = static_cast<wl_curve_ctx*>(vctx);
    hipLaunchKernelGGL(k_gauss_sum<double>, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)c->stream,
                       (const double*)c->t, (const double*)c->data, x, y, m, (int)n);
}
void wl_gauss_sum_f_s(void* vctx, size_t m, size_t n, const float* x, float* y)
{
    auto* c = static_cast<wl_curve_ctx*>(vctx);
    hipLaunchKernelGGL(k_gauss_sum<float>, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)c->stream,
                       (const float*)c->t, (const float*)c->data, x, y, m, (int)n);
}
void wl_gauss_sum_fb_d(void* vctx, size_t m, size_t n, size_t p, const double* X, double* Y)
{
    auto* c = static_cast<wl_curve_ctx*>(vctx);
    hipLaunchKernelGGL((k_gauss_sum_batched<double, false, false>), dim3(blocks_for(m * p)), dim3(256), 0, (hipStream_t)c->stream,
                       (const double*)c->t, (const double*)c->data, X, Y, m, (int)n, (int)p);
}
void wl_gauss_sum_fbr_d(void* vctx, size_t m, size_t n, size_t p, const double* X, double* Y)
{
    auto* c = static_cast<wl_curve_ctx*>(vctx);
    // a point per thread when the thread count of the grid is a multiple of p (cfg 2: p = 32 points, 256-thread workgroups)
    const unsigned nb = blocks_for(m * p);
    if ((n - 1) / 3 <= (size_t)kGaussMax && p > 0 && ((size_t)nb * 256) % p == 0)
        hipLaunchKernelGGL((k_gauss_sum_batched<double, true, true>), dim3(nb), dim3(256), 0, (hipStream_t)c->stream,
                           (const double*)c->t, (const double*)c->data, X, Y, m, (int)n, (int)p);
    else
        hipLaunchKernelGGL((k_gauss_sum_batched<double, true, false>), dim3(nb), dim3(256), 0, (hipStream_t)c->stream,
                           (const double*)c->t, (const double*)c->data, X, Y, m, (int)n, (int)p);
}
void wl_exp_decay_f_d(void* vctx, size_t m, size_t n, const double* x, double* y)
{
    (void)n;
    auto* c = static_cast<wl_curve_ctx*>(vctx);
    hipLaunchKernelGGL(k_exp_decay<double>, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)c->stream,
                       (const double*)c->t, (const double*)c->data, x, y, m, c->kind);
}
void wl_exp_decay_f_s(void* vctx, size_t m, size_t n, const float* x, float* y)
{
    (void)n;
    auto* c = static_cast<wl_curve_ctx*>(vctx);
    hipLaunchKernelGGL(k_exp_decay<float>, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)c->stream,
                       (const float*)c->t, (const float*)c->data, x, y, m, c->kind);
}

// Test helper: `blocks` workgroups of 256 threads, each holding `lds_bytes` of LDS, that do nothing but stay on their CU for
// `microseconds` (bounded: every wave leaves when the clock says so) -- the "other tenant" of the contention tests.
void wl_busy(void* stream, unsigned blocks, unsigned lds_bytes, unsigned microseconds)
{
    hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (long long)microseconds * 100);
}

// host: out[k] = u(seed + offset + k)
void wl_uniform(uint64_t seed, uint64_t offset, size_t count, double* out)
{
    for (size_t k = 0; k < count; ++k) out[k] = (double)(splitmix64_mix(seed + offset + (uint64_t)k) >> 11) * 0x1p-53;
}

// host: the tanh-linear data set of SURVEY.md 8d for rows [row_offset, row_offset + m):
//   A_ij = (2u(10; i n + j) - 1) sqrt(3/n),  x* = 2u(11) - 1,  b = tanh(A x*) + noise (2u(12; i) - 1),
//   x0 = x* + 0.1 (2u(13) - 1).  A: m*n, b: m, xstar: n, x0: n (host buffers).
void wl_tanh_linear_generate(size_t m, size_t n, size_t row_offset, double noise, double* A, double* b,
                             double* xstar, double* x0)
{
    const double sc = std::sqrt(3.0 / (double)n);
    for (size_t j = 0; j < n; ++j) {
        xstar[j] = 2 * ((double)(splitmix64_mix(11 + j) >> 11) * 0x1p-53) - 1;
        x0[j] = xstar[j] + 0.1 * (2 * ((double)(splitmix64_mix(13 + j) >> 11) * 0x1p-53) - 1);
    }
#pragma omp parallel for schedule(static)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)m; ++i) {
        const uint64_t gi = (uint64_t)row_offset + (uint64_t)i;
        double* a = A + (size_t)i * n;
        double s = 0;
        for (size_t j = 0; j < n; ++j) {
            a[j] = (2 * ((double)(splitmix64_mix(10 + gi * n + j) >> 11) * 0x1p-53) - 1) * sc;
            s += a[j] * xstar[j];
        }
        b[i] = std::tanh(s) + noise * (2 * ((double)(splitmix64_mix(12 + gi) >> 11) * 0x1p-53) - 1);
    }
}

}  // extern "C"
