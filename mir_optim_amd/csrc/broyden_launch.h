// broyden_launch.h -- host view of the read-only Broyden sweep (broyden_lr.h): argument block and launch entry points
// (defined in launch_broyden.hip, the translation unit that instantiates the kernels).
// Reference operations replaced: least_squares.d:1003-1006 (axpy, gemv, scal, ger), 1052 (gemv), 1065 (syrk) of a Broyden pass.
#pragma once

#include "common.h"
#include "solve_types.h"

namespace mirlsq {

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
    const int32_t* guard;   // optional: the kernel does nothing when *guard == 0 (rounds enqueued ahead of time)
};

// workgroups of the sweep: 4 per CU, at least ~8 row steps per wave
inline int lr_blocks(size_t m, int num_cu)
{
    const size_t G = (m + 3) / 4;
    size_t want = (G + 4 * 8 - 1) / (4 * 8);
    const size_t cap = (size_t)num_cu * 4;
    if (want > cap) want = cap;
    return (int)(want ? want : 1);
}

// the sweep: per-workgroup partial vectors -> a.partials (nblk x lr_len(n)); column a.k of U is written
template <typename T> hipError_t lr_sweep(const LrArgs<T>& a, int nblk, hipStream_t s);
// fixed-order sum of the partial vectors -> out (lr_len(n)): the all-reduce payload of a Broyden pass
template <typename T> hipError_t lr_reduce(const T* partials, int nblk, int n, T* out, const int32_t* guard, hipStream_t s);
// the n x n side: JJ += v dx^T + dx v^T + uu dx dx^T, Jy, |Jy|_inf, D_k = dx from the (all-reduced) vector lr
template <typename T> hipError_t lr_finish(const T* lr, T* D, const T* dx, int k, int n, T* JJ, T* Jy, LmState<T>* st, const int32_t* guard, hipStream_t s);
// J += the k pending terms, in place, in update order (the reference's successive `ger`s, LS:1006)
template <typename T> hipError_t lr_flush(T* J, const T* U, const T* D, int k, size_t m, int n, int num_cu, hipStream_t s);

}  // namespace mirlsq
