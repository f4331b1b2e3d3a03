// common.h -- shared device/host helpers for the MI355X LM solver (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <utility>

namespace mirlsq {

constexpr int kWave = 64;  // CDNA wavefront

#define MIRLSQ_HIP_CHECK(expr)                                                                  \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            std::fprintf(stderr, "[mir_optim_amd] HIP error %s at %s:%d: %s\n",                 \
                         hipGetErrorName(_e), __FILE__, __LINE__, #expr);                       \
            return _e;                                                                          \
        }                                                                                       \
    } while (0)

template <typename T> struct Lim;
template <> struct Lim<double> {
    static constexpr double eps = DBL_EPSILON, max = DBL_MAX, min_normal = DBL_MIN;
    __host__ __device__ static double inf() { return __builtin_huge_val(); }
};
template <> struct Lim<float> {
    static constexpr float eps = FLT_EPSILON, max = FLT_MAX, min_normal = FLT_MIN;
    __host__ __device__ static float inf() { return __builtin_huge_valf(); }
};

// ---- MFMA 16x16x4 for f64 / f32: identical A/B operand lane maps
//      (lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]);
//      the C/D maps differ: f64 row = (l >> 4) + 4 r, f32 row = 4 (l >> 4) + r; col = l & 15.
template <typename T> struct Mma;
template <> struct Mma<double> {
    using Acc = __attribute__((ext_vector_type(4))) double;
    __device__ static inline Acc mma(double a, double b, Acc c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    __host__ __device__ static inline int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mma<float> {
    using Acc = __attribute__((ext_vector_type(4))) float;
    __device__ static inline Acc mma(float a, float b, Acc c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __host__ __device__ static inline int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

template <typename T> __device__ inline T wave_shfl_xor(T v, int mask) { return __shfl_xor(v, mask, kWave); }

// rotate within each row of 16 lanes by N (DPP row_ror:N) -- a VALU lane crossbar move, ~10x
// cheaper in latency than ds_bpermute (which is what __shfl_xor lowers to)
template <int N> __device__ inline int dpp_row_ror(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, false);
}
template <int N> __device__ inline double dpp_row_ror(double v)
{
    const int lo = dpp_row_ror<N>(__double2loint(v));
    const int hi = dpp_row_ror<N>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int N> __device__ inline float dpp_row_ror(float v)
{
    return __int_as_float(dpp_row_ror<N>(__float_as_int(v)));
}

// broadcast lane N of every row of 16 lanes to the whole row (DPP row_newbcast:N, gfx90a+): a plain VALU move, no
// SGPR round trip like v_readlane and no LDS access like ds_bpermute
template <int N> __device__ inline int dpp_row_bcast(int v)
{
    // bound_ctrl = 1: no "old" operand to initialise (the source lane always exists inside the row)
    return __builtin_amdgcn_mov_dpp(v, 0x150 + N, 0xF, 0xF, true);
}
template <int N> __device__ inline double dpp_row_bcast(double v)
{
    const int lo = dpp_row_bcast<N>(__double2loint(v));
    const int hi = dpp_row_bcast<N>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int N> __device__ inline float dpp_row_bcast(float v)
{
    return __int_as_float(dpp_row_bcast<N>(__float_as_int(v)));
}

// compile-time loop: f(IntC<0>{}), ..., f(IntC<N - 1>{}) -- for builtins that need constant operands (DPP controls)
template <int V> struct IntC { static constexpr int value = V; };
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IntC<Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// sum over the 16 lanes that share lane >> 4 (every lane receives the total): rotate-and-add
template <typename T> __device__ inline T sum16(T v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    return v;
}
template <typename T> __device__ inline T wave_sum(T v) {
    v = sum16(v);
    v += wave_shfl_xor(v, 16);
    v += wave_shfl_xor(v, 32);
    return v;
}
template <typename T> __device__ inline T max16(T v) {
    T o = dpp_row_ror<8>(v); v = o > v ? o : v;
    o = dpp_row_ror<4>(v); v = o > v ? o : v;
    o = dpp_row_ror<2>(v); v = o > v ? o : v;
    o = dpp_row_ror<1>(v); v = o > v ? o : v;
    return v;
}
template <typename T> __device__ inline T wave_max(T v) {
    v = max16(v);
    T o = wave_shfl_xor(v, 16); v = o > v ? o : v;
    o = wave_shfl_xor(v, 32); v = o > v ? o : v;
    return v;
}
template <typename T> __device__ inline T wave_min(T v) { return -wave_max(-v); }

// broadcast lane `src` (wave-uniform) of v to every lane through SGPRs (v_readlane), not through LDS
__device__ inline double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ inline float lane_bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// 1/sqrt(a) and sqrt(a) to full precision from the hardware rsq estimate + Newton steps (shorter
// dependent chain than sqrt() followed by a division; used on the Cholesky pivot path)
__device__ inline void rsqrt_sqrt(double a, double& rinv, double& d)
{
    double r = __builtin_amdgcn_rsq(a);
    r = r * (1.5 - 0.5 * a * r * r);
    r = r * (1.5 - 0.5 * a * r * r);
    double s = a * r;
    s = fma(0.5 * r, fma(-s, s, a), s);          // sqrt: one correction step
    r = fma(r, fma(-s, r, 1.0), r);              // 1/sqrt consistent with s
    rinv = r; d = s;
}
__device__ inline void rsqrt_sqrt(float a, float& rinv, float& d)
{
    d = sqrtf(a);
    rinv = 1.0f / d;
}

__device__ inline double dsqrt(double v) { return sqrt(v); }
__device__ inline float dsqrt(float v) { return sqrtf(v); }
__device__ inline double dabs(double v) { return fabs(v); }
__device__ inline float dabs(float v) { return fabsf(v); }
__device__ inline double dfmax(double a, double b) { return fmax(a, b); }
__device__ inline float dfmax(float a, float b) { return fmaxf(a, b); }
__device__ inline double dfmin(double a, double b) { return fmin(a, b); }
__device__ inline float dfmin(float a, float b) { return fminf(a, b); }

}  // namespace mirlsq
