// misc_kernels.h -- the m-vector and bookkeeping kernels of the LM pass.
//
// Replaces (/root/reference/source/mir/optim/least_squares.d):
//   LS:955, LS:1115   dot(y, y)                          -> k_sumsq_partial + k_sumsq_final
//   LS:1018-1048      finite-difference column fill      -> k_fd_points + k_fd_fill / k_fd_fill_col
//   LS:1053           |Jy[iamax(Jy)]|                     -> k_unpack_grad
//   LS:1080-1161      QP/NaN guards, step guard, trial acceptance, rho, lambda/mu over a chain of
//                     speculative trials                                -> k_decide_chain
#pragma once

#include "broyden_lr.h"
#include "common.h"
#include "solve_types.h"

namespace mirlsq {

// ---- sum of squares, deterministic two-stage. Stage 1: gridDim.x partials.
// blockIdx.y selects one of several m-vectors (stride vstride) and its block of pstride partials.
template <typename T>
__global__ __launch_bounds__(256) void k_sumsq_partial(const T* __restrict__ v0, size_t m, T* __restrict__ partials0,
                                                       size_t vstride = 0, int pstride = 0)
{
    __shared__ T red[4];
    const T* __restrict__ v = v0 + (size_t)blockIdx.y * vstride;
    T* __restrict__ partials = partials0 + (size_t)blockIdx.y * pstride;
    T s = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // contiguous chunk per block keeps the summation order independent of the grid-stride pattern
    const size_t per = (m + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per;
    const size_t b1 = b0 + per < m ? b0 + per : m;
    (void)stride;
    for (size_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) { const T t = v[i]; s += t * t; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Stage 2: one block of 256 threads sums the partials in a fixed order. Collective; thread 0 returns the sum. Also run by
// k_decide_chain itself when there is no all-reduce between the two stages: same order, same bits.
template <typename T>
__device__ inline T sumsq_final_block(const T* partials, int nparts, T* red /* 4 */)
{
    T s = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const T tot = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();                                         // red may be reused
    return tot;
}
template <typename T>
__global__ __launch_bounds__(256) void k_sumsq_final(const T* __restrict__ partials0, int nparts, T* __restrict__ out, int pstride = 0)
{
    __shared__ T red[4];
    const T tot = sumsq_final_block(partials0 + (size_t)blockIdx.x * pstride, nparts, red);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// ---- LS:953-971: state at entry. *sum = ||f(x0)||^2 (already all-reduced).
// Publish the state to its mirror in pinned, device-mapped host memory: the image first, then -- behind a system-scope
// fence -- its sequence number, which the host polls instead of waiting on the stream (a stream synchronisation costs a
// copy-engine round trip plus an interrupt wake-up, tens of microseconds per decision point; a posted write over PCIe a few).
template <typename T>
__device__ inline void publish_state(const LmState<T>& s, LmState<T>* host_st, uint32_t seq)
{
    if (!host_st) return;
    LmState<T> t = s;
    t.seq = 0;
    *host_st = t;
    __threadfence_system();
    *reinterpret_cast<volatile uint32_t*>(&host_st->seq) = seq;
    __threadfence_system();
}

template <typename T>
__global__ void k_init_state(const T* sum, LmState<T>* st, LmState<T>* host_st, uint32_t seq)
{
    LmState<T> s{};
    s.lambda = 0;            // LS:966 (no warm start, quirk Q11)
    s.mu = 1;                // LS:969
    s.residual = *sum;       // LS:955
    *st = s;
    publish_state(s, host_st, seq);
}

// ---- packed [JJ lower | Jy] -> full symmetric JJ, Jy, ||Jy||_inf (LS:1053).
//      grid = n + 1 blocks: block i < n expands row i of JJ, block n handles Jy and its max.
template <typename T>
__global__ __launch_bounds__(256) void k_unpack_grad(const T* __restrict__ packed, int n, T* __restrict__ JJ,
                                                     T* __restrict__ Jy, LmState<T>* st)
{
    __shared__ T red[4];
    const int i = blockIdx.x;
    if (i < n) {
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const int r = i >= j ? i : j, c = i >= j ? j : i;
            JJ[(size_t)i * n + j] = packed[(size_t)r * (r + 1) / 2 + c];
        }
        return;
    }
    T mx = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const T v = packed[(size_t)n * (n + 1) / 2 + j];
        Jy[j] = v;
        const T av = dabs(v);
        if (av > mx) mx = av;
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        T r = red[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = red[w] > r ? red[w] : r;
        st->jy_inf = r;
    }
}

// ---- finite-difference points, LS:1027-1031: X[2j] = x with x_j = min(x_j + eps, u_j),
//      X[2j+1] = x with x_j = max(x_j - eps, l_j); twh[j] = xph - xmh.
template <typename T>
__global__ void k_fd_points(const T* __restrict__ x, const T* __restrict__ lower, const T* __restrict__ upper,
                            T eps, int n, T* __restrict__ X, T* __restrict__ twh)
{
    const int j = blockIdx.x;
    const T save = x[j];
    T xmh = save - eps, xph = save + eps;
    xmh = dfmax(xmh, lower[j]);
    xph = dfmin(xph, upper[j]);
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const T xv = x[k];
        X[(size_t)(2 * j) * n + k] = k == j ? xph : xv;
        X[(size_t)(2 * j + 1) * n + k] = k == j ? xmh : xv;
    }
    if (threadIdx.x == 0) twh[j] = xph - xmh;
}

// ---- LS:1037-1046 for a panel of columns [j0, j0 + pc): J[i][j] = (Y[2(j-j0)][i] - Y[2(j-j0)+1][i]) * (1 / twh[j])
//      (zero when twh == 0). Y rows are m-vectors, row stride ldy. LDS transpose so that both the
//      reads (along i) and the writes (along j) are coalesced. grid = (ceil(m/64), ceil(pc/32)).
template <typename T>
__global__ __launch_bounds__(256) void k_fd_fill(const T* __restrict__ Y, size_t ldy, const T* __restrict__ twh,
                                                 T* __restrict__ J, size_t m, int n, int j0, int pc)
{
    __shared__ T tile[32][65];
    const size_t i0 = (size_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63, cw = threadIdx.x >> 6;
    for (int c = cw; c < 32; c += 4) {
        const int cc = c0 + c;
        T v = 0;
        if (cc < pc && i0 + lane < m) {
            const T t = twh[j0 + cc];
            if (t != 0) {
                const T a = Y[(size_t)(2 * cc) * ldy + i0 + lane];
                const T b = Y[(size_t)(2 * cc + 1) * ldy + i0 + lane];
                T d = a;          // copy(mBuffer, Jj)
                d += T(-1) * b;   // axpy(-1, mBuffer, Jj)
                v = d * (T(1) / t);   // scal(1 / twh, Jj)
            }
        }
        tile[c][lane] = v;
    }
    __syncthreads();
    const int ncol = (pc - c0) < 32 ? (pc - c0) : 32;
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
        const int r = idx >> 5, c = idx & 31;
        if (c < ncol && i0 + r < m) J[(i0 + r) * (size_t)n + j0 + c0 + c] = tile[c][r];
    }
}

// single strided column (host-callback mode): J[:, j] = (yp - ym) * (1/twh) or 0
template <typename T>
__global__ __launch_bounds__(256) void k_fd_fill_col(const T* __restrict__ yp, const T* __restrict__ ym, T twh,
                                                     T* __restrict__ J, size_t m, int n, int j)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    T v = 0;
    if (twh != 0) { T d = yp[i]; d += T(-1) * ym[i]; v = d * (T(1) / twh); }
    J[i * (size_t)n + j] = v;
}

// ---- LS:1080-1161 for a chain of ks speculative trials: walks the lambda ladder in the reference's order --
//      QP failure / NaN step (LS:1080-1092), step-size guard (LS:1101-1106), trial residual (LS:1117), rejection
//      (LS:1125-1130) -- until the first accepted trial, whose x, dx and scalars become the state (LS:1132-1161).
//      sums[k] = ||f(trial_k)||^2 (already all-reduced). Later chain entries are simply discarded: they were
//      computed on the assumption that every earlier entry is rejected, which is exactly when the reference would
//      have computed them. One block of kSolveThreads.
// Collective over the workgroup (any size >= kReduceRanges threads when nparts > 0); returns LmState::spec_ok as it was
// decided -- 1: the fused round's kernel goes on with the Broyden finish and the next solve (solve_kernel.h).
// One trial and no more parameters than threads (the fused round): entry 0 of the trial point and of the step, loaded by the
// caller BEFORE the decision is known, so that their memory latency runs beside the decision's own loads.
template <typename T> struct DecidePre { T xv, dv; };

template <typename T>
__device__ inline int decide_chain_body(const DecideArgs<T>& a, const DecidePre<T>* pre = nullptr, long long* dbg = nullptr)
{
    __shared__ int acc_s;
    __shared__ LmState<T> s_pub;
    if (a.nparts > 0) {
        __shared__ T part[kReduceRanges];
        for (int k = 0; k < a.ks; ++k) {
            const T tot = lr_reduce_scalar(a.partials + (size_t)k * a.pstride, a.nparts, part);
            if (threadIdx.x == 0) a.sums[k] = tot;           // read back by thread 0 below (and by the host's trace)
        }
    }
    if (threadIdx.x == 0) {
        LmState<T> s = *a.st;
        if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[29] = wall_clock64(); }
        int dec = kDecideReject, acc = -1;
        uint32_t consumed = 0, fcalls = 0, rejects = 0, guards = 0, qpact = 0, null_tail = 0;
        for (int k = 0; k < a.ks; ++k) {
            const ChainRec<T> r = a.rec[k];
            if (k == 0 && a.check_grad && (r.flags & kFlagGradSmall)) { dec = kDecideGradSmall; break; }   // LS:1053
            ++consumed;
            if (k == 0 && a.lambda_from_state) s.lambda = r.lambda;           // lambda_0, LS:1067-1072
            if (r.qp_iterations > 0) ++qpact;
            s.qp_status = r.qp_status; s.qp_iterations = r.qp_iterations; s.flags = r.flags;
            if (r.qp_status != 0 || (r.flags & kFlagDxNaN)) { dec = kDecideNumericError; break; }   // LS:1080-1092
            if (r.flags & kFlagStepTooLong) {                                 // LS:1101-1106
                s.lambda *= a.set.lambdaIncrease * s.mu;
                s.mu *= 2;
                ++guards;
                continue;
            }
            ++fcalls;                                                         // LS:1112
            // null step: trial == x, the pure callback would return the residual vector already held (kFlagNullStep)
            const bool null_step = (r.flags & kFlagNullStep) != 0;
            if (null_step) a.sums[k] = s.residual;
            null_tail = null_step ? 1u : 0u;
            const T tr = a.sums[k];
            s.trial_residual = tr;
            if (!(tr <= Lim<T>::inf())) { dec = kDecideNumericError; s.flags |= kFlagTrialNotFinite; break; }   // LS:1117
            const T improvement = s.residual - tr;                            // LS:1124
            s.improvement = improvement;
            if (!(improvement > 0)) {                                         // LS:1125-1130
                s.lambda *= a.set.lambdaIncrease * s.mu;
                s.mu *= 2;
                ++rejects;
                continue;
            }
            s.mu = 1;                                                         // LS:1132-1139
            s.iterations++;
            s.residual = tr;
            s.dx_dot = r.new_dx_dot;
            s.new_dx_dot = r.new_dx_dot;
            s.predicted = r.predicted;
            s.trial_xnorm = r.trial_xnorm;
            acc = k;
            if (!(r.predicted > 0)) { dec = kDecideAcceptNoPrediction; break; }   // LS:1144-1148
            const T rho = r.predicted / improvement;                          // LS:1150 (quirk Q2)
            s.rho = rho;
            if (rho < a.set.minStepQuality) {                                 // LS:1152-1156
                s.lambda *= a.set.lambdaIncrease * s.mu;
                s.mu *= 2;
            } else if (rho >= a.set.goodStepQuality) {                        // LS:1158-1161
                s.lambda = dfmax(a.set.lambdaDecrease * s.lambda * s.mu, a.set.minLambda);
            }
            dec = kDecideAccept;
            break;
        }
        // Fused round (spec_static): may the kernel go on with the Broyden finish and the next solve? Only when this round
        // ended in a plain acceptance of its one trial and none of the reference's exit / refresh tests fires before the next
        // pass (LS:974, 979, 990, 1144, 1164-1173, 1175; lambda >= minLambda: no lambda_0 rule, LS:1067): then the next pass is
        // a Broyden update + one solve -- the sweep behind the trial residual has prepared exactly that (lr_spec_go, the same
        // expression on the same record, says whether it ran in full).
        int spec = 0;
        if (a.spec_static && dec == kDecideAccept && acc == 0) {
            spec = !(s.residual <= a.set.maxGoodResidual) && s.iterations < a.maxIterations && (s.lambda <= a.set.maxLambda)
                && (s.lambda >= a.set.minLambda) && lr_spec_go(a.rec[0], a.set.absTolerance, a.set.relTolerance);
        }
        s.spec_ok = spec;
        uint32_t rescued = 0;
        for (int k = 0; k < a.ks; ++k) if (a.rec[k].flags & kFlagCoopRescued) ++rescued;
        s.coop_rescued = rescued;
        s.decision = dec; s.accepted_k = acc; s.consumed = consumed; s.fcalls = fcalls;
        s.rejects = rejects; s.guards = guards; s.qp_active = qpact;
        s.null_tail = (dec == kDecideReject) ? null_tail : 0u;
        *a.st = s;
        s_pub = s;
        acc_s = acc;
        // the image of the state goes to the pinned mirror now (sequence number 0: not valid yet), beside the accepted point below
        if (a.host_st) { LmState<T> t = s; t.seq = 0; *a.host_st = t; }
    }
    lds_barrier();                       // (acc_s, s_pub; thread 0's stores to memory are drained by the barriers further down)
    const int acc = acc_s;
    if (acc >= 0) {
        if (pre) {                                                            // (acc == 0: there is one trial)
            if ((int)threadIdx.x < a.n) {
                a.x[threadIdx.x] = pre->xv;                                   // LS:1135
                if (a.host_x) a.host_x[threadIdx.x] = pre->xv;
                a.dx_acc[threadIdx.x] = pre->dv;
            }
        } else {
            for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
                const T xv = a.trial[(size_t)acc * a.n + i];
                a.x[i] = xv;                                                  // LS:1135
                if (a.host_x) a.host_x[i] = xv;
                a.dx_acc[i] = a.dx_chain[(size_t)acc * a.n + i];
            }
        }
    }
    if (dbg && threadIdx.x == 0) dbg[30] = wall_clock64();
    if (a.host_st) {
        // ONE system-scope fence per thread: the accepted point and the image are in host memory before the sequence number
        // that announces them (the host polls that word; nothing waits for it here)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *reinterpret_cast<volatile uint32_t*>(&a.host_st->seq) = a.seq;
    }
    if (dbg && threadIdx.x == 0) dbg[31] = wall_clock64();
    const int spec = s_pub.spec_ok;
    __syncthreads();                     // x, dx_acc and the state are visible to the whole workgroup; s_pub may be reused
    return spec;
}

template <typename T>
__global__ __launch_bounds__(kSolveThreads) void k_decide_chain(DecideArgs<T> a)
{
    (void)decide_chain_body(a);
}

// ---- LS:984-989: forced refresh resets mu
template <typename T>
__global__ void k_reset_mu(LmState<T>* st) { st->mu = 1; }

}  // namespace mirlsq
