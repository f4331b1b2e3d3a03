// launch_broyden.hip -- the translation unit that instantiates the read-only Broyden sweep and its companions
// (broyden_lr.h) and defines the launch entry points declared in broyden_launch.h.
#include <hip/hip_runtime.h>

#include "broyden_launch.h"
#include "launch_util.h"
#include "broyden_lr.h"

namespace mirlsq {

namespace {
template <typename T, int NCP>
hipError_t lr_sweep_ncp(const LrArgs<T>& a, int nblk, bool vec, hipStream_t s)
{
    if (vec) MIRLSQ_LAUNCH((k_broyden_lr<T, NCP, true>), dim3(nblk), dim3(256), 0, s, a);
    else MIRLSQ_LAUNCH((k_broyden_lr<T, NCP, false>), dim3(nblk), dim3(256), 0, s, a);
    return hipGetLastError();
}
}  // namespace
template <typename T>
hipError_t lr_sweep(const LrArgs<T>& a, int nblk, hipStream_t s)
{
    if (a.n > kLrMaxN || a.k < 0 || a.k >= kLrMax) return hipErrorInvalidValue;
    const bool vec = (a.n % 2 == 0) && (reinterpret_cast<uintptr_t>(a.J) % (2 * sizeof(T)) == 0);
    const int ncp = (a.n + 31) / 32;
    if (ncp <= 1) return lr_sweep_ncp<T, 1>(a, nblk, vec, s);
    if (ncp <= 2) return lr_sweep_ncp<T, 2>(a, nblk, vec, s);
    if (ncp <= 4) return lr_sweep_ncp<T, 4>(a, nblk, vec, s);
    if (ncp <= 8) return lr_sweep_ncp<T, 8>(a, nblk, vec, s);
    return lr_sweep_ncp<T, 16>(a, nblk, vec, s);            // 256 < n <= 512: one workgroup a CU by registers, still one sweep over J
}

namespace {
template <typename T, int NCP>
hipError_t lr_flush_ncp(T* J, const T* U, const T* D, int k, size_t m, int n, int nblk, bool vec, hipStream_t s)
{
    const size_t lds = (size_t)k * n * sizeof(T);
    if (vec) MIRLSQ_LAUNCH((k_lr_flush<T, NCP, true>), dim3(nblk), dim3(256), lds, s, J, U, D, k, m, n);
    else MIRLSQ_LAUNCH((k_lr_flush<T, NCP, false>), dim3(nblk), dim3(256), lds, s, J, U, D, k, m, n);
    return hipGetLastError();
}
}  // namespace
template <typename T>
hipError_t lr_flush(T* J, const T* U, const T* D, int k, size_t m, int n, int num_cu, hipStream_t s)
{
    if (k <= 0) return hipSuccess;
    if (n > kLrMaxN || k > kLrMax) return hipErrorInvalidValue;
    const size_t G = (m + 3) / 4;
    size_t blocks = (G + 3) / 4;
    if (blocks > (size_t)num_cu * 8) blocks = (size_t)num_cu * 8;
    if (blocks < 1) blocks = 1;
    const bool vec = (n % 2 == 0) && (reinterpret_cast<uintptr_t>(J) % (2 * sizeof(T)) == 0);
    const int ncp = (n + 31) / 32;
    if (ncp <= 1) return lr_flush_ncp<T, 1>(J, U, D, k, m, n, (int)blocks, vec, s);
    if (ncp <= 2) return lr_flush_ncp<T, 2>(J, U, D, k, m, n, (int)blocks, vec, s);
    if (ncp <= 4) return lr_flush_ncp<T, 4>(J, U, D, k, m, n, (int)blocks, vec, s);
    if (ncp <= 8) return lr_flush_ncp<T, 8>(J, U, D, k, m, n, (int)blocks, vec, s);
    return lr_flush_ncp<T, 16>(J, U, D, k, m, n, (int)blocks, vec, s);
}

template <typename T>
hipError_t lr_reduce(const T* partials, int nblk, int n, T* out, hipStream_t s)
{
    const int len = lr_len(n);
    MIRLSQ_LAUNCH(k_lr_reduce<T>, dim3((len + 31) / 32), dim3(32 * kReduceRanges), 0, s, partials, nblk, len, out);
    return hipGetLastError();
}

template <typename T>
hipError_t lr_finish(const T* lr, T* D, const T* dx, int k, int n, T* JJ, T* Jy, LmState<T>* st, hipStream_t s)
{
    MIRLSQ_LAUNCH(k_lr_finish<T>, dim3(n + 1), dim3(256), 0, s, lr, D, dx, k, n, JJ, Jy, st);
    return hipGetLastError();
}

template <typename T>
hipError_t lr_sumsq(const T* v, size_t m, int count, size_t vstride, T* partials, int pstride, int nblk, hipStream_t s)
{
    if (count < 1 || nblk < 1 || nblk > pstride) return hipErrorInvalidValue;
    MIRLSQ_LAUNCH(k_lr_sumsq<T>, dim3(nblk, count), dim3(256), 0, s, v, m, vstride, partials, pstride);
    return hipGetLastError();
}
template <typename T>
hipError_t lr_sumsq_final(const T* partials, int pstride, int nblk, int count, T* out, hipStream_t s)
{
    MIRLSQ_LAUNCH(k_lr_sumsq_final<T>, dim3(count), dim3(256), 0, s, partials, pstride, nblk, out);
    return hipGetLastError();
}

#define MIRLSQ_INSTANTIATE(T)                                                                                             \
    template hipError_t lr_sweep<T>(const LrArgs<T>&, int, hipStream_t);                                                  \
    template hipError_t lr_reduce<T>(const T*, int, int, T*, hipStream_t);                                                \
    template hipError_t lr_finish<T>(const T*, T*, const T*, int, int, T*, T*, LmState<T>*, hipStream_t);                 \
    template hipError_t lr_sumsq<T>(const T*, size_t, int, size_t, T*, int, int, hipStream_t);                            \
    template hipError_t lr_sumsq_final<T>(const T*, int, int, int, T*, hipStream_t);                                      \
    template hipError_t lr_flush<T>(T*, const T*, const T*, int, size_t, int, int, hipStream_t);
MIRLSQ_INSTANTIATE(double)
MIRLSQ_INSTANTIATE(float)
#undef MIRLSQ_INSTANTIATE

}  // namespace mirlsq

MIRLSQ_DEFINE_PRELOAD(broyden)
