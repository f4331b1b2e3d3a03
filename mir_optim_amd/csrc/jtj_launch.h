// jtj_launch.h -- host side of the J^T J kernels: which kernel runs for a shape (JtjPlan) and how it is launched.
//
// Product path (variant == 0), LS = /root/reference/source/mir/optim/least_squares.d:
//   f64, n <= 128               k_jtj_fdp<NCB, true>    finite-difference panel -> J, J^T J, J^T y   (LS:1041-1047, 1052, 1065)
//   f64, n <= 128, n even       k_jtj_fdp<NCB, false>   J^T J + J^T y of a given J                   (LS:1052, 1065)
//   f64, 128 < n <= 256 (n % 16 == 0, m even)  k_jtj8   eight-wave LDS-DMA ring
//   everything else             k_jtj (n <= 128) / k_jtj_wide (any n: 64-column tile pairs)
// The `variant` bits of mir_lsq_gpu_options select the literal restatements the tests compare the product path with.
// Nothing here is process-global state except the per-device "attribute set" masks, which are atomics.
#pragma once

#include <atomic>

#include "../../include/mir_optim_amd.h"
#include "common.h"
#include "jtj_fdp.h"
#include "jtj_fdp8.h"
#include "jtj_kernel.h"
#include "jtj_pc32.h"
#include "jtj_ring8.h"
#include "jtj_wide.h"

namespace mirlsq {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: one bit per device ordinal records
// that it has been raised there (devices >= 64 simply set it every time). Safe under concurrent solves.
inline hipError_t ensure_dyn_lds(const void* fn, size_t bytes, std::atomic<uint64_t>& done)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 64;
    const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}
#define MIRLSQ_ENSURE_LDS(kern, bytes)                                                            \
    do {                                                                                          \
        static std::atomic<uint64_t> lds_done_{0};                                                \
        const hipError_t e_ = ensure_dyn_lds(reinterpret_cast<const void*>(kern), (bytes), lds_done_); \
        if (e_ != hipSuccess) return e_;                                                          \
    } while (0)

// Where the slab reduction may put its result besides `packed`: the full symmetric J^T J and J^T y (the work of
// k_unpack_grad; max |J^T y| is taken by the solve kernel), when no all-reduce of `packed` sits in between. All null: `packed` only.
template <typename T>
struct JtjUnpack {
    T* JJ = nullptr; T* Jy = nullptr;
};

struct JtjPlan {
    int ncb = 0;
    int nblk = 0;
    int slab_len = 0;
    size_t lds = 0;
    bool v2 = false;        // LDS-DMA ring kernel k_jtj2 (f64, n = 16 ncb <= 128, m even): MIR_LSQ_VARIANT_JTJ_RING / BROYDEN_REWRITE
    bool wide = false;      // n > 128 and not ring8: 64-column tile-pair jobs (jtj_wide.h), any n
    bool ring8 = false;     // 128 < n <= 256, f64, n % 16 == 0, m even: eight-wave LDS-DMA ring (jtj_ring8.h)
    bool fdp = false;       // f64, n <= 128, any m: producer / consumer kernel (jtj_fdp.h) for the finite-difference J^T J
    bool fdp_plain = false; // ... and, n even, for the plain J^T J
    bool fdp8 = false;      // f64, 128 < n <= 256, n % 32 == 0, any m: eight producer + consumer waves (jtj_fdp8.h), FD J^T J only
    int fdp8_nblk = 0, fdp8_slab_len = 0;
    bool pc32 = false;      // f32, n <= 128, n % 4 == 0, any m: producer / consumer kernel on v_mfma_f32_16x16x4 (jtj_pc32.h), plain J^T J
    int pc32_nblk = 0;
    int njobs = 1;
};

inline size_t jtj2_lds_rt(int ncb, bool br)
{
    switch (ncb) {
    case 1: return br ? Jtj2Cfg<1, true>::LDS_BYTES : Jtj2Cfg<1, false>::LDS_BYTES;
    case 2: return br ? Jtj2Cfg<2, true>::LDS_BYTES : Jtj2Cfg<2, false>::LDS_BYTES;
    case 3: return br ? Jtj2Cfg<3, true>::LDS_BYTES : Jtj2Cfg<3, false>::LDS_BYTES;
    case 4: return br ? Jtj2Cfg<4, true>::LDS_BYTES : Jtj2Cfg<4, false>::LDS_BYTES;
    case 5: return br ? Jtj2Cfg<5, true>::LDS_BYTES : Jtj2Cfg<5, false>::LDS_BYTES;
    case 6: return br ? Jtj2Cfg<6, true>::LDS_BYTES : Jtj2Cfg<6, false>::LDS_BYTES;
    case 7: return br ? Jtj2Cfg<7, true>::LDS_BYTES : Jtj2Cfg<7, false>::LDS_BYTES;
    case 8: return br ? Jtj2Cfg<8, true>::LDS_BYTES : Jtj2Cfg<8, false>::LDS_BYTES;
    }
    return 0;
}
inline int jtj2_rs_rt(int ncb)
{
    switch (ncb) {
    case 1: return Jtj2Cfg<1, false>::RS; case 2: return Jtj2Cfg<2, false>::RS; case 3: return Jtj2Cfg<3, false>::RS;
    case 4: return Jtj2Cfg<4, false>::RS; case 5: return Jtj2Cfg<5, false>::RS; case 6: return Jtj2Cfg<6, false>::RS;
    case 7: return Jtj2Cfg<7, false>::RS; case 8: return Jtj2Cfg<8, false>::RS;
    }
    return 4;
}
inline size_t jtj8_lds_rt(int ncb)
{
    switch (ncb) {
    case 9: return Jtj8Cfg<9>::LDS_BYTES; case 10: return Jtj8Cfg<10>::LDS_BYTES; case 11: return Jtj8Cfg<11>::LDS_BYTES;
    case 12: return Jtj8Cfg<12>::LDS_BYTES; case 13: return Jtj8Cfg<13>::LDS_BYTES; case 14: return Jtj8Cfg<14>::LDS_BYTES;
    case 15: return Jtj8Cfg<15>::LDS_BYTES; case 16: return Jtj8Cfg<16>::LDS_BYTES;
    }
    return 0;
}

template <typename T>
JtjPlan jtj_plan(size_t m, int n, int num_cu, uint32_t variant = 0)
{
    JtjPlan p;
    p.ncb = (n + 15) / 16;
    const int nacc = p.ncb * (p.ncb + 1) / 2;
    p.slab_len = (nacc * 4 + p.ncb) * kWave;
    const bool stream = (variant & MIR_LSQ_VARIANT_JTJ_STREAM) != 0;
    if (sizeof(T) == 8 && n > 128 && n <= 256 && n % 32 == 0 && !stream) {
        p.fdp8 = true;
        const size_t stot = (m + 15) / 16;
        const size_t want = (stot + 7) / 8;                    // at least ~8 stages per workgroup
        p.fdp8_nblk = (int)(want < (size_t)num_cu ? (want ? want : 1) : (size_t)num_cu);   // one workgroup per CU
        p.fdp8_slab_len = p.slab_len;
    }
    if (sizeof(T) == 8 && n > 128 && n <= 256 && n % 16 == 0 && m % 2 == 0 && !stream) {
        p.ring8 = true;
        p.lds = jtj8_lds_rt(p.ncb);
        const size_t stot = (m + 15) / 16;
        size_t want = (stot + 7) / 8;                          // at least ~8 stages per workgroup
        p.nblk = (int)(want < (size_t)num_cu ? (want ? want : 1) : (size_t)num_cu);   // one workgroup per CU
        return p;
    }
    if (n > 128) {
        p.wide = true;
        const int nt = (p.ncb + kWideTile - 1) / kWideTile;
        p.njobs = nt * (nt + 1) / 2;
        p.slab_len = kWideSlabLen;
        p.lds = (size_t)2 * kWideSlabLen * sizeof(T);
        const size_t G = (m + 3) / 4;
        size_t want = (G + 4 * 8 - 1) / (4 * 8);
        size_t cap = (size_t)num_cu * 4 / p.njobs;
        if (cap < 1) cap = 1;
        p.nblk = (int)(want < cap ? (want ? want : 1) : cap);
        return p;
    }
    if (sizeof(T) == 4 && n <= 128 && n % 4 == 0 && !stream) {
        p.pc32 = true;
        const size_t stot = (m + 63) / 64;                     // 64-row stages
        const size_t want = (stot + 3) / 4;                    // at least ~4 stages per workgroup
        const size_t cap = (size_t)num_cu * 2;
        p.pc32_nblk = (int)(want < cap ? (want ? want : 1) : cap);
    }
    p.fdp = sizeof(T) == 8 && n <= 128 && !stream;
    p.fdp_plain = p.fdp && n % 2 == 0;
    if (sizeof(T) == 8 && n % 16 == 0 && n <= 128 && m % 2 == 0 && !stream) {
        p.v2 = true;
        p.lds = jtj2_lds_rt(p.ncb, false);
        const size_t stot = (m + jtj2_rs_rt(p.ncb) - 1) / jtj2_rs_rt(p.ncb);
        size_t want = (stot + 7) / 8;                          // at least ~8 stages per workgroup
        const size_t cap = (size_t)num_cu * 2;
        p.nblk = (int)(want < cap ? (want ? want : 1) : cap);
        return p;
    }
    const int rpb = 4 * (int)(sizeof(T) / 4);
    const int roles = jtj_roles_rt(p.ncb, rpb);
    p.lds = (size_t)(roles == 4 ? 0 : (roles == 2 ? 1 : 2)) * p.slab_len * sizeof(T);
    // workgroups per CU: LDS- and register-limited (one workgroup = one wave per SIMD)
    int per_cu = p.lds ? (int)((160 * 1024) / p.lds) : 8;
    const int reg_waves = (nacc * rpb / roles > 40) ? 2 : 4;   // matches jtj_min_waves
    if (per_cu > reg_waves) per_cu = reg_waves;
    if (per_cu < 1) per_cu = 1;
    const size_t G = (m + 3) / 4;
    const size_t slots_per_blk = kJtjWaves / roles;
    size_t want = (G + slots_per_blk * 8 - 1) / (slots_per_blk * 8);     // at least ~8 row groups per wave
    size_t cap = (size_t)num_cu * per_cu;
    p.nblk = (int)(want < cap ? (want ? want : 1) : cap);
    return p;
}

// ---- slab reduction shared by every one-job kernel: -> packed[ n(n+1)/2 + n ]
template <typename T>
inline hipError_t jtj_reduce_slabs(const JtjPlan& p, const JtjArgs<T>& a, T* packed, hipStream_t s, const JtjUnpack<T>& u = {},
                                   int nslabs = -1, int slab_len = -1)
{
    if (nslabs < 0) { nslabs = p.nblk; slab_len = p.slab_len; }
    const int rb = (slab_len + 31) / 32;
    MIRLSQ_LAUNCH(k_jtj_slab_reduce<T>, dim3(rb), dim3(1024), 0, s, a.slabs, nslabs, slab_len, p.ncb, a.n, packed,
                  u.JJ, u.Jy);
    return hipGetLastError();
}
// does jtj_run honour a JtjUnpack for this plan? (the tile-pair jobs have a reduction of their own; jtj_run_fd* always do)
inline bool jtj_plain_unpacks(const JtjPlan& p) { return !p.wide; }

// ---- k_jtj: register streaming (f32, odd n; MIR_LSQ_VARIANT_JTJ_STREAM)
template <typename T, int NCB, bool BR>
hipError_t jtj_stream_one(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    auto kern = k_jtj<T, NCB, BR>;
    MIRLSQ_ENSURE_LDS(kern, p.lds);
    MIRLSQ_LAUNCH(kern, dim3(p.nblk), dim3(256), p.lds, s, a);
    return hipGetLastError();
}
template <typename T, bool BR>
hipError_t jtj_stream(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    switch (p.ncb) {
    case 1: return jtj_stream_one<T, 1, BR>(p, a, s);
    case 2: return jtj_stream_one<T, 2, BR>(p, a, s);
    case 3: return jtj_stream_one<T, 3, BR>(p, a, s);
    case 4: return jtj_stream_one<T, 4, BR>(p, a, s);
    case 5: return jtj_stream_one<T, 5, BR>(p, a, s);
    case 6: return jtj_stream_one<T, 6, BR>(p, a, s);
    case 7: return jtj_stream_one<T, 7, BR>(p, a, s);
    case 8: return jtj_stream_one<T, 8, BR>(p, a, s);
    }
    return hipErrorInvalidValue;
}

// ---- k_jtj_fdp: producer / consumer waves (jtj_fdp.h); FD = the finite-difference panel is the source
template <int NCB, bool FD, bool DIFF = false>
hipError_t jtj_fdp_one(const JtjPlan& p, const JtjArgs<double>& a, hipStream_t s)
{
    using FC = JtjFdpCfg<NCB, FD>;
    MIRLSQ_ENSURE_LDS((k_jtj_fdp<NCB, FD, DIFF>), (size_t)FC::LDS_BYTES);
    MIRLSQ_LAUNCH((k_jtj_fdp<NCB, FD, DIFF>), dim3(p.nblk), dim3(FC::THREADS), FC::LDS_BYTES, s, a);
    return hipGetLastError();
}
template <typename T, bool FD, bool DIFF = false>
hipError_t jtj_fdp_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) {
        switch (p.ncb) {
        case 1: return jtj_fdp_one<1, FD, DIFF>(p, a, s);
        case 2: return jtj_fdp_one<2, FD, DIFF>(p, a, s);
        case 3: return jtj_fdp_one<3, FD, DIFF>(p, a, s);
        case 4: return jtj_fdp_one<4, FD, DIFF>(p, a, s);
        case 5: return jtj_fdp_one<5, FD, DIFF>(p, a, s);
        case 6: return jtj_fdp_one<6, FD, DIFF>(p, a, s);
        case 7: return jtj_fdp_one<7, FD, DIFF>(p, a, s);
        case 8: return jtj_fdp_one<8, FD, DIFF>(p, a, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- k_jtj_pc32: the f32 producer / consumer kernel (jtj_pc32.h)
template <int NCB>
hipError_t jtj_pc32_one(const JtjPlan& p, const JtjArgs<float>& a, hipStream_t s)
{
    using C = JtjPc32Cfg<NCB>;
    MIRLSQ_ENSURE_LDS((k_jtj_pc32<NCB>), (size_t)C::LDS_BYTES);
    MIRLSQ_LAUNCH((k_jtj_pc32<NCB>), dim3(p.pc32_nblk), dim3(C::THREADS), C::LDS_BYTES, s, a);
    return hipGetLastError();
}
template <typename T>
hipError_t jtj_pc32_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    if constexpr (sizeof(T) == 4) {
        switch (p.ncb) {
        case 1: return jtj_pc32_one<1>(p, a, s);
        case 2: return jtj_pc32_one<2>(p, a, s);
        case 3: return jtj_pc32_one<3>(p, a, s);
        case 4: return jtj_pc32_one<4>(p, a, s);
        case 5: return jtj_pc32_one<5>(p, a, s);
        case 6: return jtj_pc32_one<6>(p, a, s);
        case 7: return jtj_pc32_one<7>(p, a, s);
        case 8: return jtj_pc32_one<8>(p, a, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- k_jtj2: LDS-DMA ring; BR = Broyden update fused in, J rewritten (the literal restatement of LS:1003-1006)
template <int NCB, bool BR>
hipError_t jtj2_one(const JtjPlan& p, const JtjArgs<double>& a, hipStream_t s)
{
    auto kern = k_jtj2<NCB, BR>;
    constexpr size_t lds = Jtj2Cfg<NCB, BR>::LDS_BYTES;
    MIRLSQ_ENSURE_LDS(kern, lds);
    MIRLSQ_LAUNCH(kern, dim3(p.nblk), dim3(kJtj2Threads), lds, s, a);
    return hipGetLastError();
}
template <typename T, bool BR>
hipError_t jtj2_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) {
        switch (p.ncb) {
        case 1: return jtj2_one<1, BR>(p, a, s);
        case 2: return jtj2_one<2, BR>(p, a, s);
        case 3: return jtj2_one<3, BR>(p, a, s);
        case 4: return jtj2_one<4, BR>(p, a, s);
        case 5: return jtj2_one<5, BR>(p, a, s);
        case 6: return jtj2_one<6, BR>(p, a, s);
        case 7: return jtj2_one<7, BR>(p, a, s);
        case 8: return jtj2_one<8, BR>(p, a, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- k_jtj8: eight-wave ring, 128 < n <= 256
template <int NCB>
hipError_t jtj8_one(const JtjPlan& p, const JtjArgs<double>& a, bool broyden, hipStream_t s)
{
    auto kern = k_jtj8<NCB>;
    constexpr size_t lds = Jtj8Cfg<NCB>::LDS_BYTES;
    MIRLSQ_ENSURE_LDS(kern, lds);
    MIRLSQ_LAUNCH(kern, dim3(p.nblk), dim3(kJtj8Threads), lds, s, a, broyden ? 1 : 0);
    return hipGetLastError();
}
template <typename T>
hipError_t jtj8_launch(const JtjPlan& p, const JtjArgs<T>& a, bool broyden, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) {
        switch (p.ncb) {
        case 9: return jtj8_one<9>(p, a, broyden, s);
        case 10: return jtj8_one<10>(p, a, broyden, s);
        case 11: return jtj8_one<11>(p, a, broyden, s);
        case 12: return jtj8_one<12>(p, a, broyden, s);
        case 13: return jtj8_one<13>(p, a, broyden, s);
        case 14: return jtj8_one<14>(p, a, broyden, s);
        case 15: return jtj8_one<15>(p, a, broyden, s);
        case 16: return jtj8_one<16>(p, a, broyden, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- tile-pair jobs, any n > 128 (jtj_wide.h); the Broyden rewrite is a separate pass in front
template <typename T>
hipError_t jtj_run_wide(const JtjPlan& p, const JtjArgs<T>& a, bool broyden, T* packed, hipStream_t s)
{
    if (broyden) {
        const size_t G = (a.m + 3) / 4;
        size_t blocks = (G + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        if (a.n <= 256)
            MIRLSQ_LAUNCH(k_broyden_wide<T>, dim3((unsigned)blocks), dim3(256), 0, s, a.Jout, a.y, a.y_old, a.dx, a.dx_dot, a.m, a.n);
        else
            MIRLSQ_LAUNCH(k_broyden_rows<T>, dim3((unsigned)blocks), dim3(256), 0, s, a.Jout, a.y, a.y_old, a.dx, a.dx_dot, a.m, a.n);
    }
    JtjWideArgs<T> w{};
    w.J = a.J; w.y = a.y; w.slabs = a.slabs; w.m = a.m; w.n = a.n;
    w.nt = ((a.n + 15) / 16 + kWideTile - 1) / kWideTile;
    MIRLSQ_ENSURE_LDS(k_jtj_wide<T>, p.lds);
    MIRLSQ_LAUNCH(k_jtj_wide<T>, dim3(p.nblk, p.njobs), dim3(256), p.lds, s, w);
    MIRLSQ_LAUNCH(k_jtj_wide_reduce<T>, dim3((kWideSlabLen + 31) / 32, p.njobs), dim3(256), 0, s, a.slabs, p.nblk, a.n, packed);
    return hipGetLastError();
}

// ---- [Broyden rewrite] + J^T J + J^T y -> packed[ n(n+1)/2 + n ]
template <typename T>
hipError_t jtj_run(const JtjPlan& p, const JtjArgs<T>& a, bool broyden, T* packed, hipStream_t s, uint32_t variant = 0,
                   const JtjUnpack<T>& u = {})
{
    hipError_t e;
    if (p.pc32 && !broyden) {
        e = jtj_pc32_launch<T>(p, a, s);
        if (e != hipSuccess) return e;
        return jtj_reduce_slabs<T>(p, a, packed, s, u, p.pc32_nblk, p.slab_len);
    }
    if (p.ring8) e = jtj8_launch<T>(p, a, broyden, s);
    else if (p.wide) return jtj_run_wide<T>(p, a, broyden, packed, s);
    else if (broyden) e = p.v2 ? jtj2_launch<T, true>(p, a, s) : jtj_stream<T, true>(p, a, s);
    else if (p.fdp_plain && !((variant & MIR_LSQ_VARIANT_JTJ_RING) && p.v2)) e = jtj_fdp_launch<T, false>(p, a, s);
    else if (p.v2) e = jtj2_launch<T, false>(p, a, s);
    else e = jtj_stream<T, false>(p, a, s);
    if (e != hipSuccess) return e;
    return jtj_reduce_slabs<T>(p, a, packed, s, u);
}

// ---- k_jtj_fdp8: the finite-difference J^T J for 128 < n <= 256 (jtj_fdp8.h)
template <int NCB, bool DIFF = false>
hipError_t jtj_fdp8_one(const JtjPlan& p, const JtjArgs<double>& a, hipStream_t s)
{
    using FC = JtjFdp8Cfg<NCB>;
    MIRLSQ_ENSURE_LDS((k_jtj_fdp8<NCB, DIFF>), (size_t)FC::LDS_BYTES);
    MIRLSQ_LAUNCH((k_jtj_fdp8<NCB, DIFF>), dim3(p.fdp8_nblk), dim3(FC::THREADS), FC::LDS_BYTES, s, a);
    return hipGetLastError();
}
template <typename T, bool DIFF = false>
hipError_t jtj_fdp8_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) {
        if constexpr (DIFF) {                              // two columns per 16-byte load: whole loads per row need n % 64 == 0
            switch (p.ncb) {
            case 12: return jtj_fdp8_one<12, true>(p, a, s);
            case 16: return jtj_fdp8_one<16, true>(p, a, s);
            }
        } else {
            switch (p.ncb) {
            case 10: return jtj_fdp8_one<10>(p, a, s);
            case 12: return jtj_fdp8_one<12>(p, a, s);
            case 14: return jtj_fdp8_one<14>(p, a, s);
            case 16: return jtj_fdp8_one<16>(p, a, s);
            }
        }
    }
    return hipErrorInvalidValue;
}

// ---- finite-difference panel (a.J: m x 2n row-major, a.twh) -> a.Jout, packed
template <typename T>
hipError_t jtj_run_fd(const JtjPlan& p, const JtjArgs<T>& a, T* packed, hipStream_t s, const JtjUnpack<T>& u = {})
{
    if (p.fdp8) {
        const hipError_t e = jtj_fdp8_launch<T>(p, a, s);
        if (e != hipSuccess) return e;
        return jtj_reduce_slabs<T>(p, a, packed, s, u, p.fdp8_nblk, p.fdp8_slab_len);
    }
    if (!p.fdp) return hipErrorInvalidValue;
    const hipError_t e = jtj_fdp_launch<T, true>(p, a, s);
    if (e != hipSuccess) return e;
    return jtj_reduce_slabs<T>(p, a, packed, s, u);
}

// ---- finite-difference DIFFERENCE panel (a.J: m x n row-major, D_ij = f(x + h e_j)_i - f(x - h e_j)_i; a.twh) -> a.Jout, packed
//      (f64; n <= 128, n even: JtjPlan::fdp_plain; n = 192, 256: k_jtj_fdp8)
inline bool jtj_fd_diff_ok(const JtjPlan& p, int n) { return p.fdp_plain || (p.fdp8 && n % 64 == 0); }
template <typename T>
hipError_t jtj_run_fd_diff(const JtjPlan& p, const JtjArgs<T>& a, T* packed, hipStream_t s, const JtjUnpack<T>& u = {})
{
    if (p.fdp8 && a.n % 64 == 0) {
        const hipError_t e = jtj_fdp8_launch<T, true>(p, a, s);
        if (e != hipSuccess) return e;
        return jtj_reduce_slabs<T>(p, a, packed, s, u, p.fdp8_nblk, p.fdp8_slab_len);
    }
    if (!p.fdp_plain) return hipErrorInvalidValue;
    const hipError_t e = jtj_fdp_launch<T, false, true>(p, a, s);
    if (e != hipSuccess) return e;
    return jtj_reduce_slabs<T>(p, a, packed, s, u);
}

}  // namespace mirlsq
