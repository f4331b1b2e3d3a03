// launch_jtj.hip -- the ONE translation unit that instantiates the J^T J kernels (jtj_kernel.h, jtj_fdp.h, jtj_fdp8.h,
// jtj_pc32.h, jtj_ring8.h, jtj_wide.h) and defines the launch entry points declared in jtj_plan.h.
// Reference operations replaced: least_squares.d:1052 (gemv J^T y), 1065 (syrk J^T J), 1041-1047 (finite-difference column
// arithmetic, fused), and -- MIR_LSQ_VARIANT_BROYDEN_REWRITE only -- 1003-1006 (Broyden update with J rewritten).
#include <hip/hip_runtime.h>

#include "jtj_plan.h"
#include "launch_util.h"
#include "solve_types.h"
#include "jtj_kernel.h"
#include "jtj_fdp.h"
#include "jtj_fdp8.h"
#include "jtj_pc32.h"
#include "jtj_ring8.h"
#include "jtj_wide.h"
#include "misc_kernels.h"

namespace mirlsq {

static_assert(jtj8_lds_bytes(9) == Jtj8Cfg<9>::LDS_BYTES && jtj8_lds_bytes(16) == Jtj8Cfg<16>::LDS_BYTES, "jtj_plan.h: k_jtj8 LDS size");

namespace {

// ---- slab reduction shared by every one-job kernel: -> packed[ n(n+1)/2 + n ]
template <typename T>
hipError_t reduce_slabs(const JtjPlan& p, const JtjArgs<T>& a, T* packed, hipStream_t s, const JtjUnpack<T>& u, int nslabs, int slab_len,
                        int ncb = 0)
{
    // ncb: the block count the slabs were laid out for (a kernel compiled for a padded n: > p.ncb; rows / columns >= n are skipped)
    const int rb = (slab_len + 31) / 32;
    MIRLSQ_LAUNCH(k_jtj_slab_reduce<T>, dim3(rb), dim3(1024), 0, s, a.slabs, nslabs, slab_len, ncb ? ncb : p.ncb, a.n, packed, u.JJ, u.Jy);
    return hipGetLastError();
}

// ---- k_jtj: register streaming (f32 with n % 4 != 0, f64 with odd n; BR: the Broyden rewrite)
template <typename T, int NCB, bool BR>
hipError_t stream_one(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    auto kern = k_jtj<T, NCB, BR>;
    MIRLSQ_ENSURE_LDS(kern, p.stream_lds);
    MIRLSQ_LAUNCH(kern, dim3(p.stream_nblk), dim3(256), p.stream_lds, s, a);
    return hipGetLastError();
}
template <typename T, bool BR>
hipError_t stream_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    switch (p.ncb) {
    case 1: return stream_one<T, 1, BR>(p, a, s);
    case 2: return stream_one<T, 2, BR>(p, a, s);
    case 3: return stream_one<T, 3, BR>(p, a, s);
    case 4: return stream_one<T, 4, BR>(p, a, s);
    case 5: return stream_one<T, 5, BR>(p, a, s);
    case 6: return stream_one<T, 6, BR>(p, a, s);
    case 7: return stream_one<T, 7, BR>(p, a, s);
    case 8: return stream_one<T, 8, BR>(p, a, s);
    }
    return hipErrorInvalidValue;
}

// ---- k_jtj_fdp: producer / consumer waves (jtj_fdp.h); FD = the finite-difference panel is the source
template <int NCB, bool FD, bool DIFF>
hipError_t fdp_one(const JtjPlan& p, const JtjArgs<double>& a, hipStream_t s)
{
    using FC = JtjFdpCfg<NCB, FD>;
    if constexpr (!FD) {
        // plain J / difference panel: rows of n doubles. Odd n, or a source that is not 16-byte aligned (an offset view handed to
        // a unit entry): the flat producer (the same 16-byte loads, 8-byte aligned, over the wave's contiguous rows)
        // ... and the difference panel whenever a row of J does not start on a 128-byte boundary (n % 16 != 0): the flat
        // producer writes J back in memory order instead of the consumers' row segments. (On the 128-byte grid the consumers'
        // write-back is the faster one: 0.428 against 0.448 ms at m = 1e6, n = 128; 0.189 against 0.196 at n = 64.)
        if (a.n % 2 != 0 || reinterpret_cast<uintptr_t>(a.J) % 16 != 0 || (DIFF && a.n % 16 != 0)) {
            constexpr size_t lds = (size_t)FC::LDS_BYTES + FC::N * sizeof(double);       // + the 1 / twh table of the flat producer
            MIRLSQ_ENSURE_LDS((k_jtj_fdp<NCB, FD, DIFF, true>), lds);
            MIRLSQ_LAUNCH((k_jtj_fdp<NCB, FD, DIFF, true>), dim3(p.nblk), dim3(FC::THREADS), lds, s, a);
            return hipGetLastError();
        }
    }
    MIRLSQ_ENSURE_LDS((k_jtj_fdp<NCB, FD, DIFF>), (size_t)FC::LDS_BYTES);
    MIRLSQ_LAUNCH((k_jtj_fdp<NCB, FD, DIFF>), dim3(p.nblk), dim3(FC::THREADS), FC::LDS_BYTES, s, a);
    return hipGetLastError();
}
template <typename T, bool FD, bool DIFF = false>
hipError_t fdp_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) {
        switch (p.ncb) {
        case 1: return fdp_one<1, FD, DIFF>(p, a, s);
        case 2: return fdp_one<2, FD, DIFF>(p, a, s);
        case 3: return fdp_one<3, FD, DIFF>(p, a, s);
        case 4: return fdp_one<4, FD, DIFF>(p, a, s);
        case 5: return fdp_one<5, FD, DIFF>(p, a, s);
        case 6: return fdp_one<6, FD, DIFF>(p, a, s);
        case 7: return fdp_one<7, FD, DIFF>(p, a, s);
        case 8: return fdp_one<8, FD, DIFF>(p, a, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- k_jtj_pc32: the f32 producer / consumer kernel (jtj_pc32.h). It reads J with 16-byte loads of four floats: a J that
//      is not 16-byte aligned (an offset view handed to the unit entry mir_lsq_jtj_s) takes the register-streaming kernel
template <int NCB>
hipError_t pc32_one(const JtjPlan& p, const JtjArgs<float>& a, hipStream_t s)
{
    using C = JtjPc32Cfg<NCB>;
    MIRLSQ_ENSURE_LDS((k_jtj_pc32<NCB>), (size_t)C::LDS_BYTES);
    MIRLSQ_LAUNCH((k_jtj_pc32<NCB>), dim3(p.pc32_nblk), dim3(C::THREADS), C::LDS_BYTES, s, a);
    return hipGetLastError();
}
template <typename T>
hipError_t pc32_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    if constexpr (sizeof(T) == 4) {
        switch (p.ncb) {
        case 1: return pc32_one<1>(p, a, s);
        case 2: return pc32_one<2>(p, a, s);
        case 3: return pc32_one<3>(p, a, s);
        case 4: return pc32_one<4>(p, a, s);
        case 5: return pc32_one<5>(p, a, s);
        case 6: return pc32_one<6>(p, a, s);
        case 7: return pc32_one<7>(p, a, s);
        case 8: return pc32_one<8>(p, a, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- k_jtj8: eight-wave ring, 128 < n <= 256
template <int NCB>
hipError_t ring8_one(const JtjPlan& p, const JtjArgs<double>& a, bool broyden, hipStream_t s)
{
    auto kern = k_jtj8<NCB>;
    constexpr size_t lds = Jtj8Cfg<NCB>::LDS_BYTES;
    MIRLSQ_ENSURE_LDS(kern, lds);
    MIRLSQ_LAUNCH(kern, dim3(p.nblk), dim3(kJtj8Threads), lds, s, a, broyden ? 1 : 0);
    return hipGetLastError();
}
template <typename T>
hipError_t ring8_launch(const JtjPlan& p, const JtjArgs<T>& a, bool broyden, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) {
        switch (p.ncb) {
        case 9: return ring8_one<9>(p, a, broyden, s);
        case 10: return ring8_one<10>(p, a, broyden, s);
        case 11: return ring8_one<11>(p, a, broyden, s);
        case 12: return ring8_one<12>(p, a, broyden, s);
        case 13: return ring8_one<13>(p, a, broyden, s);
        case 14: return ring8_one<14>(p, a, broyden, s);
        case 15: return ring8_one<15>(p, a, broyden, s);
        case 16: return ring8_one<16>(p, a, broyden, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- tile-pair jobs, any n > 128 (jtj_wide.h); the Broyden rewrite is a separate pass in front
template <typename T>
hipError_t run_wide(const JtjPlan& p, const JtjArgs<T>& a, bool broyden, T* packed, hipStream_t s)
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

// ---- k_jtj_fdp8: the finite-difference J^T J for 128 < n <= 256 (jtj_fdp8.h)
template <int NCB, bool DIFF = false, bool PLAIN = false>
hipError_t fdp8_one(const JtjPlan& p, const JtjArgs<double>& a, hipStream_t s)
{
    using FC = JtjFdp8Cfg<NCB>;
    MIRLSQ_ENSURE_LDS((k_jtj_fdp8<NCB, DIFF, PLAIN>), (size_t)FC::LDS_BYTES);
    MIRLSQ_LAUNCH((k_jtj_fdp8<NCB, DIFF, PLAIN>), dim3(p.fdp8_nblk), dim3(FC::THREADS), FC::LDS_BYTES, s, a);
    return hipGetLastError();
}
// the plain J^T J of a given J for the shapes the eight-wave ring does not take (n % 16 != 0, odd m)
template <typename T>
hipError_t fdp8_plain_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) {
        switch (p.fdp8_ncb) {
        case 10: return fdp8_one<10, false, true>(p, a, s);
        case 12: return fdp8_one<12, false, true>(p, a, s);
        case 14: return fdp8_one<14, false, true>(p, a, s);
        case 16: return fdp8_one<16, false, true>(p, a, s);
        }
    }
    return hipErrorInvalidValue;
}
template <typename T, bool DIFF = false>
hipError_t fdp8_launch(const JtjPlan& p, const JtjArgs<T>& a, hipStream_t s)
{
    if constexpr (sizeof(T) == 8) {
        if constexpr (DIFF) {                              // two columns per 16-byte load: whole loads per row need n % 64 == 0
            if (a.n % 64 != 0) return hipErrorInvalidValue;
            switch (p.fdp8_ncb) {
            case 12: return fdp8_one<12, true>(p, a, s);
            case 16: return fdp8_one<16, true>(p, a, s);
            }
        } else {
            switch (p.fdp8_ncb) {                          // the pair panel: any n, compiled for n rounded up to a multiple of 32
            case 10: return fdp8_one<10>(p, a, s);
            case 12: return fdp8_one<12>(p, a, s);
            case 14: return fdp8_one<14>(p, a, s);
            case 16: return fdp8_one<16>(p, a, s);
            }
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace

template <typename T>
hipError_t jtj_run(const JtjPlan& p, const JtjArgs<T>& a, bool broyden, T* packed, hipStream_t s, const JtjUnpack<T>& u)
{
    hipError_t e;
    if (p.pc32 && !broyden && reinterpret_cast<uintptr_t>(a.J) % 16 == 0) {
        e = pc32_launch<T>(p, a, s);
        if (e != hipSuccess) return e;
        return reduce_slabs<T>(p, a, packed, s, u, p.pc32_nblk, p.slab_len);
    }
    if (p.ring8) {
        e = ring8_launch<T>(p, a, broyden, s);
        if (e != hipSuccess) return e;
        return reduce_slabs<T>(p, a, packed, s, u, p.nblk, p.slab_len);
    }
    if (p.wide && p.fdp8 && !broyden) {                    // 128 < n <= 256 off the ring's grid: the plain flavour of k_jtj_fdp8
        e = fdp8_plain_launch<T>(p, a, s);
        if (e != hipSuccess) return e;
        return reduce_slabs<T>(p, a, packed, s, u, p.fdp8_nblk, p.fdp8_slab_len, p.fdp8_ncb);
    }
    if (p.wide) return run_wide<T>(p, a, broyden, packed, s);
    if (!broyden && p.fdp_plain) {
        e = fdp_launch<T, false>(p, a, s);
        if (e != hipSuccess) return e;
        return reduce_slabs<T>(p, a, packed, s, u, p.nblk, p.slab_len);
    }
    e = broyden ? stream_launch<T, true>(p, a, s) : stream_launch<T, false>(p, a, s);
    if (e != hipSuccess) return e;
    return reduce_slabs<T>(p, a, packed, s, u, p.stream_nblk, p.slab_len);
}

template <typename T>
hipError_t jtj_run_fd(const JtjPlan& p, const JtjArgs<T>& a, T* packed, hipStream_t s, const JtjUnpack<T>& u)
{
    if (p.fdp8) {
        const hipError_t e = fdp8_launch<T>(p, a, s);
        if (e != hipSuccess) return e;
        return reduce_slabs<T>(p, a, packed, s, u, p.fdp8_nblk, p.fdp8_slab_len, p.fdp8_ncb);
    }
    if (!p.fdp) return hipErrorInvalidValue;
    const hipError_t e = fdp_launch<T, true>(p, a, s);
    if (e != hipSuccess) return e;
    return reduce_slabs<T>(p, a, packed, s, u, p.nblk, p.slab_len);
}

template <typename T>
hipError_t jtj_run_fd_diff(const JtjPlan& p, const JtjArgs<T>& a, T* packed, hipStream_t s, const JtjUnpack<T>& u)
{
    if (p.fdp8 && a.n % 64 == 0) {
        const hipError_t e = fdp8_launch<T, true>(p, a, s);
        if (e != hipSuccess) return e;
        return reduce_slabs<T>(p, a, packed, s, u, p.fdp8_nblk, p.fdp8_slab_len, p.fdp8_ncb);
    }
    if (!p.fdp_plain) return hipErrorInvalidValue;
    const hipError_t e = fdp_launch<T, false, true>(p, a, s);
    if (e != hipSuccess) return e;
    return reduce_slabs<T>(p, a, packed, s, u, p.nblk, p.slab_len);
}

template <typename T>
hipError_t jtj_unpack(const T* packed, int n, T* JJ, T* Jy, LmState<T>* st, hipStream_t s)
{
    MIRLSQ_LAUNCH(k_unpack_grad<T>, dim3(n + 1), dim3(128), 0, s, packed, n, JJ, Jy, st);
    return hipGetLastError();
}

#define MIRLSQ_INSTANTIATE(T)                                                                                                              \
    template hipError_t jtj_run<T>(const JtjPlan&, const JtjArgs<T>&, bool, T*, hipStream_t, const JtjUnpack<T>&);                         \
    template hipError_t jtj_run_fd<T>(const JtjPlan&, const JtjArgs<T>&, T*, hipStream_t, const JtjUnpack<T>&);                            \
    template hipError_t jtj_run_fd_diff<T>(const JtjPlan&, const JtjArgs<T>&, T*, hipStream_t, const JtjUnpack<T>&);                       \
    template hipError_t jtj_unpack<T>(const T*, int, T*, T*, LmState<T>*, hipStream_t);
MIRLSQ_INSTANTIATE(double)
MIRLSQ_INSTANTIATE(float)
#undef MIRLSQ_INSTANTIATE

}  // namespace mirlsq

MIRLSQ_DEFINE_PRELOAD(jtj)
