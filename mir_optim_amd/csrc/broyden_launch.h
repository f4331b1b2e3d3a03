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
    // SPECULATIVE sweep (the fused round of solver_loop.hip): run right behind the trial residual y = f(trial), before the trial
    // is decided, on the assumption that it will be accepted -- dx / dx_dot are the ladder entry's rounded step and its
    // record's new_dx_dot. spec_rec != nullptr: the record of that entry; when it already says that no Broyden pass can follow
    // (QP failure, NaN, step guard, null step, gradient test, or the x-convergence test LS:1164-1173 that forces a refresh) the
    // kernel only forms ||y||^2 (entry lr_yy(n)) and zeros. Either way entry lr_yy(n) carries the bits k_lr_sumsq would produce.
    const ChainRec<T>* spec_rec;
    T absTolerance, relTolerance;
};

// workgroups of the sweep: 4 per CU, at least ~8 row steps per wave (never more than kLrMaxBlocks: the trial sums' partials
// of k_lr_sumsq share the layout)
constexpr int kLrMaxBlocks = 1024;
inline int lr_blocks(size_t m, int num_cu)
{
    const size_t G = (m + 3) / 4;
    size_t want = (G + 4 * 8 - 1) / (4 * 8);
    size_t cap = (size_t)num_cu * 4;
    if (cap > (size_t)kLrMaxBlocks) cap = kLrMaxBlocks;
    if (want > cap) want = cap;
    return (int)(want ? want : 1);
}

// the sweep: per-workgroup partial vectors -> a.partials (nblk x lr_len(n)); column a.k of U is written
template <typename T> hipError_t lr_sweep(const LrArgs<T>& a, int nblk, hipStream_t s);
// fixed-order sum of the partial vectors -> out (lr_len(n)): the all-reduce payload of a Broyden pass
template <typename T> hipError_t lr_reduce(const T* partials, int nblk, int n, T* out, hipStream_t s);
// the n x n side: JJ += v dx^T + dx v^T + uu dx dx^T, Jy, |Jy|_inf, D_k = dx from the (all-reduced) vector lr
template <typename T> hipError_t lr_finish(const T* lr, T* D, const T* dx, int k, int n, T* JJ, T* Jy, LmState<T>* st, hipStream_t s);
// ||v_k||^2 of `count` m-vectors (v + k vstride), stage 1: nblk partials each at partials + k pstride -- the row walk and the
// order of the sums are the sweep's, so that a trial's sum of squares has the same bits whether it rode on a sweep or not
template <typename T> hipError_t lr_sumsq(const T* v, size_t m, int count, size_t vstride, T* partials, int pstride, int nblk, hipStream_t s);
// ... stage 2 (the order of lr_reduce): out[k] = the sum of the nblk partials of vector k
template <typename T> hipError_t lr_sumsq_final(const T* partials, int pstride, int nblk, int count, T* out, hipStream_t s);
// J += the k pending terms, in place, in update order (the reference's successive `ger`s, LS:1006)
template <typename T> hipError_t lr_flush(T* J, const T* U, const T* D, int k, size_t m, int n, int num_cu, hipStream_t s);

}  // namespace mirlsq
