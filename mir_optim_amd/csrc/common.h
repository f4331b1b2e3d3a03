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

// Every kernel the LIBRARY launches on behalf of a solve goes through MIRLSQ_LAUNCH: the per-thread counter is what
// mir_lsq_stats.library_launches reports (the caller's callbacks and memory copies are not counted).
inline thread_local uint64_t tl_launches = 0;
#define MIRLSQ_LAUNCH(...) do { ++::mirlsq::tl_launches; hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// one rounding, whatever the compiler's contraction setting: sums that two kernels must form bit for bit alike
__device__ inline double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline float dfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// ---- pending rank-one Broyden terms (broyden_lr.h): sizes shared by the sweep, its reduction and the n x n finish
constexpr int kLrMax = 16;
// [ v0 (n) | g0 (n) | w (kLrMax) | h (kLrMax) | uu | uy | yy ]   yy = ||y_new||^2: the trial's sum of squares (LS:1115) rides on
// the sweep that is run speculatively behind the trial residual, one all-reduce for both (lr_yy(n) = its index)
__host__ __device__ constexpr int lr_len(int n) { return 2 * n + 2 * kLrMax + 3; }
__host__ __device__ constexpr int lr_yy(int n) { return 2 * n + 2 * kLrMax + 2; }
constexpr int kLrMaxN = 512;     // widest problem of the read-only Broyden sweep (16 column-pair chunks of 32 a lane); above, J is rewritten
constexpr int kReduceRanges = 32;
// one entry of the symmetric rank-two update J_k^T J_k = J_{k-1}^T J_{k-1} + v dx^T + dx v^T + uu dx dx^T, r >= c (k_lr_finish)
// Written out in fused multiply-adds: the SAME roundings wherever the term is formed (k_lr_finish; the fused round's kernel adds
// it to the registers that hold J^T J) -- left to the compiler's contraction, two call sites of `(vr dc + dr vc) + uu dr dc` may
// round different products (seen: one entry of J^T J an ulp apart between the two, tests/test_gpu_fused_rounds.py).
template <typename T>
__device__ __forceinline__ T lr_jj_term(T vr, T vc, T dr, T dc, T uu)
{
#pragma clang fp contract(off)
    return dfma(uu * dr, dc, dfma(vr, dc, dr * vc));
}

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
    // accumulator row c lives in lane group crow_group(c), register crow_reg(c)
    // accumulator row h (lane group h & 3, register h >> 2) relabelled so that a lane group owns FOUR CONSECUTIVE indices:
    // perm(h) = 4 (h & 3) + (h >> 2) (an involution); group g, register q <-> index 4 g + q
    __host__ __device__ static constexpr int perm(int h) { return 4 * (h & 3) + (h >> 2); }
};
template <> struct Mma<float> {
    using Acc = __attribute__((ext_vector_type(4))) float;
    __device__ static inline Acc mma(float a, float b, Acc c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __host__ __device__ static inline int row(int lane, int r) { return 4 * (lane >> 4) + r; }
    __host__ __device__ static constexpr int perm(int h) { return h; }     // group g already owns rows 4 g .. 4 g + 3
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

// Lanes of ONE wave exchanging data through LDS without a workgroup barrier: the DS operations of a wave execute in order,
// but the COMPILER assumes data-race freedom -- without this it keeps a value a lane loaded earlier when only other lanes
// stored to the address since (seen: the reload sunk into the storing lanes' branch). No instruction is emitted.
__device__ inline void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS accesses only: outstanding GLOBAL stores are not waited for (a __syncthreads() drains them:
// 1-2 us when a workgroup has just written tens of values per thread that nobody in it reads back soon).
__device__ inline void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// sum over each quad of lanes (4 l .. 4 l + 3), every lane receives the total: DPP quad_perm [1,0,3,2] then [2,3,0,1]
template <int CTRL> __device__ inline int dpp_quad(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ inline double dpp_quad(double v)
{
    const int lo = dpp_quad<CTRL>(__double2loint(v));
    const int hi = dpp_quad<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ inline float dpp_quad(float v) { return __int_as_float(dpp_quad<CTRL>(__float_as_int(v))); }
template <typename T> __device__ inline T quad_sum(T v)
{
    v += dpp_quad<0xB1>(v);
    v += dpp_quad<0x4E>(v);
    return v;
}

// lane (i, k = lane >> 4) <- register k of lane (i, JB): one row of the 4 x 4 (register x lane group) transpose, on the gfx950
// swap instructions (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second;
// v_permlane32_swap: upper 32 lanes of the first <-> lower 32 lanes of the second). VALU only: no LDS round trip.
template <int JB> __device__ inline int group_pick(int r0, int r1, int r2, int r3)
{
    const auto a = __builtin_amdgcn_permlane16_swap(r0, r1, false, false);   // a[0] = r0g0 r1g0 r0g2 r1g2, a[1] = r0g1 r1g1 r0g3 r1g3
    const auto b = __builtin_amdgcn_permlane16_swap(r2, r3, false, false);
    if constexpr ((JB & 1) == 0) {
        const auto s = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);   // s[0] = r0g0 r1g0 r2g0 r3g0, s[1] = r0g2 .. r3g2
        return JB == 0 ? s[0] : s[1];
    } else {
        const auto s = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
        return JB == 1 ? s[0] : s[1];
    }
}
template <int JB> __device__ inline double group_pick(double r0, double r1, double r2, double r3)
{
    const int lo = group_pick<JB>(__double2loint(r0), __double2loint(r1), __double2loint(r2), __double2loint(r3));
    const int hi = group_pick<JB>(__double2hiint(r0), __double2hiint(r1), __double2hiint(r2), __double2hiint(r3));
    return __hiloint2double(hi, lo);
}
template <int JB> __device__ inline float group_pick(float r0, float r1, float r2, float r3)
{
    return __int_as_float(group_pick<JB>(__float_as_int(r0), __float_as_int(r1), __float_as_int(r2), __float_as_int(r3)));
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
// float: the last two levels as the classic gfx9 DPP reduction -- row_bcast:15 adds the last lane of rows 0 and 2 into rows 1
// and 3, row_bcast:31 the last lane of row 1 into rows 2 and 3 -- and lane 63, which then holds the total, is read back through
// v_readlane: a wave-uniform result in a scalar register, no trip through the LDS crossbar (ds_bpermute). The pairs of partial
// sums are those of the butterfly: same bits. (The v_permlane16/32_swap instructions do the exchange in one operation each, but
// a build with them gave wrong sums now and then, in some processes and not in others: not used here.)
template <> __device__ inline float wave_sum<float>(float v) {
    v = sum16(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <> __device__ inline double wave_sum<double>(double v) {
    v = sum16(v);
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xA, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xA, 0xF, false);
        v += __hiloint2double(hi, lo);
    }
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x143, 0xC, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x143, 0xC, 0xF, false);
        v += __hiloint2double(hi, lo);
    }
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
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
    // v_rsq_f64, one coupled (Goldschmidt) step for sqrt and 1 / (2 sqrt), two residual corrections of the root, one of the
    // reciprocal: 10 dependent operations (the chain is on the critical path of every Cholesky pivot)
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, a), h, g);
    g = fma(fma(-g, g, a), h, g);
    const double rh = h + h;
    rinv = fma(fma(-g, rh, 1.0), rh, rh);
    d = g;
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
