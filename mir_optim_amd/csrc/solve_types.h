// solve_types.h -- the plain data shared by the n x n kernels (solve_kernel.h, solve_lds.h, solve_big.h), the decision
// kernels (misc_kernels.h) and the host driver: device-resident LM state, the records of the lambda ladder, kernel
// argument blocks, and the host-side shape queries. No device code: the host translation units include only this.
#pragma once

#include "common.h"

namespace mirlsq {

constexpr int kSolveThreads = 256;
constexpr int kSolveMaxN = 256;          // n-vectors are handled one element per thread
constexpr int kSolveLdsBytes = 159 * 1024;

// Device-resident scalars of the LM loop. The host mirrors it after each decision point.
template <typename T>
struct LmState {
    T lambda, mu, residual, trial_residual;
    T dx_dot, new_dx_dot, predicted, trial_xnorm;
    T jy_inf, improvement, rho, pad0;
    int32_t qp_status, qp_iterations, flags, decision;
    uint32_t iterations;
    int32_t accepted_k;        // chain index of the accepted trial (-1: none)
    uint32_t consumed;         // chain steps the reference would have executed this round
    uint32_t fcalls;           // residual evaluations among them (LS:1112)
    uint32_t rejects, guards, qp_active;
    uint32_t null_tail;        // the last consumed trial of the round was a null step (trial == x): the next ones probably are too
    uint32_t seq;              // host mirror only: number of the decision point this image belongs to (written last)
    uint32_t coop_rescued;     // ladder entries of the round whose helper workgroups did not answer in time (solve_coop.h) and that
                               // the rescue launch solved again on one workgroup: the host counts them and stops asking for helpers
    int32_t spec_ok;           // set by the decision at the head of a fused round: the Broyden pass the round's sweep prepared is the
                               // one the reference runs next (accepted step, no exit test fired) -- the kernel went on with it
};

// One solve of the lambda ladder (chain step k): everything the acceptance logic needs about it.
constexpr int kChainMax = 8;
template <typename T>
struct ChainRec {
    T lambda, new_dx_dot, predicted, trial_xnorm;
    int32_t qp_status, qp_iterations, flags, pad;
};
// kFlagNullStep: the rounded step (LS:1096-1097) is exactly zero in every component, so trial == x bit for bit. The
// callbacks are `pure` (LS:73-80): f(trial) is the residual vector the solver already holds, ||f(trial)||^2 == residual,
// improvement == 0 and the pass is rejected (LS:1125) -- the evaluation can be elided without changing any result.
enum : int32_t { kFlagDxNaN = 1, kFlagXNaN = 2, kFlagStepTooLong = 4, kFlagTrialNotFinite = 8, kFlagGradSmall = 16,
                 kFlagNullStep = 32,
                 kFlagCoopRescued = 64 };   // the entry's helpers timed out; the rescue launch solved it again on one workgroup
// ChainRec::qp_status beyond solveBoxQP's own 0 / 1 / 2 (QP:18-26): the entry's helper workgroups did not answer within
// kCoopSpinSeconds -- a scheduling fact, not a numeric one; only ever seen between the any-n solve's launch and its rescue launch
constexpr int32_t kQpCoopTimeout = 3;
enum : int32_t {
    kDecideNone = 0, kDecideReject = 1, kDecideAccept = 2, kDecideAcceptNoPrediction = 3,
    kDecideNumericError = 4, kDecideGradSmall = 5
};

template <typename T>
struct LmSettingsDev {   // the floating-point part of LeastSquaresSettings!T (LS:85-123)
    T jacobianEpsilon, absTolerance, relTolerance, gradTolerance, maxGoodResidual, maxStep, maxLambda,
      minLambda, minStepQuality, goodStepQuality, lambdaIncrease, lambdaDecrease, qpRelTolerance, qpAbsTolerance;
    uint32_t qpMaxIterations, pad;
};

template <typename T>
struct SolveScratch {    // global scratch, all L2 resident
    T* Pm;      // n x n, P = JJ + lambda I, full symmetric (unscaled; BOXCQP reads it)
    T* A;       // n x n, the (possibly equilibrated) matrix handed to posvx, full symmetric
    T* Fg;      // n x (n|1) factor when it does not fit LDS
    T* vec;     // 12 n-vectors: s, b, r, w, la, mu, sX, qpl, qpu, q, xq, spare
    int32_t* ivec;  // 2 n: SI, flags
    long long* dbg; // optional phase stamps (diagnostic builds of the host pass a buffer; else nullptr)
    unsigned long long* coop;   // n > kSolveMaxN: the sync words of the entry's helper workgroups (solve_coop.h), zero at creation
    T* cS;                      // ... and their scratch: two n x kCoopBlockCols look-ahead blocks of ?potrf + 2 x kCoopMaxPeers n-vectors of partial products
};

// helper workgroups of the any-n solve (solve_coop.h): sync words per ladder entry (64-bit; one 128-byte line per word that is
// polled), peers per entry for a given n, and the entry's scratch behind cS (two n x kCoopBlockCols look-ahead blocks of
// kCoopPanels 16-column panels each, then -- at coop_part_offset(n) -- 2 x kCoopMaxPeers n-vectors of partial products)
constexpr int kCoopMaxPeers = 16;
constexpr int kCoopLine = 16;
constexpr int kCoopWords = kCoopLine * (kCoopMaxPeers + 3);
__host__ __device__ inline int coop_peers(int n)
{
    if (n <= kSolveMaxN) return 1;
    const int nb = (n + 15) / 16;
    const int w = (nb + 3) / 4;                     // about four 16-row blocks a peer's eight waves at the start of ?potrf
    return w < 2 ? 2 : (w > kCoopMaxPeers ? kCoopMaxPeers : w);
}
constexpr int kCoopPanels = 2;                                  // kCoopPotrfT: 16-column panels of a look-ahead block
constexpr int kCoopBlockCols = 16 * kCoopPanels;
__host__ __device__ inline size_t coop_part_offset(int n) { return (size_t)2 * n * kCoopBlockCols; }
__host__ __device__ inline size_t coop_scratch_elems(int n) { return n > kSolveMaxN ? coop_part_offset(n) + (size_t)2 * kCoopMaxPeers * n : 0; }

constexpr int kLdsBlk = 272;     // solve_lds.h: 16 x 17 elements per LDS block
__host__ __device__ constexpr int lds_solve_elems(int nb) { return nb * (nb + 1) * kLdsBlk + 48 * nb + 2; }

// fast-path tile count for n (0 = generic path with the factor in global memory) and its LDS bytes:
// factor (n|1) x 16 NB, then colbuf, rdiag (16 NB each) and one scalar
__host__ __device__ inline int solve_nb(int n, int elem)
{
    const int nb = n <= 16 ? 1 : (n <= 32 ? 2 : (n <= 64 ? 4 : (n <= 128 ? 8 : 0)));
    if (nb == 0) return 0;
    const long bytes = (long)lds_solve_elems(nb) * elem;      // L and A block triangles + three vectors (solve_lds.h)
    return bytes <= kSolveLdsBytes ? nb : 0;
}
__host__ __device__ inline size_t solve_lds_bytes(int n, int elem)
{
    const int nb = solve_nb(n, elem);
    return nb ? (size_t)lds_solve_elems(nb) * elem : 0;
}

// the decision of a round (k_decide_chain / the head of a fused round; misc_kernels.h)
template <typename T>
struct DecideArgs {
    T* sums;            // ks trial sums of squares; entries of null steps are filled in here (= the current residual)
    const ChainRec<T>* rec;
    LmState<T>* st;
    LmSettingsDev<T> set;
    T* x;               // n: current point, overwritten by the accepted trial (LS:1135)
    const T* trial;     // ks x n
    const T* dx_chain;  // ks x n
    T* dx_acc;          // n: accepted step, kept for the next Broyden update (LS:1004-1006)
    int n, ks, check_grad, lambda_from_state;
    LmState<T>* host_st;  // pinned mirror (device-mapped) or nullptr
    T* host_x;            // pinned, n: receives the accepted point
    uint32_t seq;         // sequence number of this decision point
    int spec_static;      // the decision is the head of a fused round (k_lm_solve with a.fused): decide whether the kernel goes on
    uint32_t maxIterations;
    const T* partials;    // nparts > 0: sums[k] is still the nparts stage-1 partials of k_lr_sumsq at partials + k pstride (no
    int nparts, pstride;  // all-reduce sits between the stages: single GPU) -- this kernel runs stage 2 itself (lr_reduce_scalar)
};

template <typename T>
struct LmSolveArgs {
    const T* JJ;       // n x n full symmetric, undamped
    const T* Jy;       // n
    const T* x;        // n current point
    const T* lower;    // n
    const T* upper;    // n
    T* dx;             // kChainMax x n out: rounded step (LS:1096-1097) of chain step blockIdx.x
    T* trial;          // kChainMax x n out: clamp(x + dx) (LS:1108-1110)
    LmState<T>* st;
    ChainRec<T>* rec;  // kChainMax
    LmSettingsDev<T> set;
    SolveScratch<T> sc[kChainMax];
    T lam[kChainMax];  // the lambda ladder lambda_k = lambda after k bumps (LS:1103/1127); workgroup k solves with lam[k]
    int n;
    int f_in_lds;
    int check_grad;        // a new Jy was just computed: apply the gradient test LS:1053 first (chain of 1)
    int lambda_from_state; // step 0 takes st->lambda and applies the lambda_0 rule LS:1067-1072
    int lambda_from_device; // step 0 takes st->lambda as it is (the fused round: written by the decision at the kernel's head)
    // FUSED ROUND (fused != 0; one ladder entry, n <= kSolveMaxN): the kernel first DECIDES the previous round's trial (dec;
    // its sum of squares is entry lr_yy(n) of the all-reduced sweep vector fin_lr) and publishes the decision; if that is a plain
    // acceptance after which a Broyden pass follows, it applies the pass's n x n side (k_lr_finish's arithmetic: J^T J, J^T y,
    // D[fin_k] = dx) from the rest of fin_lr and goes on with the solve -- decision, finish and solve in ONE launch.
    int fused;
    DecideArgs<T> dec;
    const T* fin_lr;
    T* fin_D;
    T* fin_JJ;
    T* fin_Jy;
    int fin_k;
    int coop_w;            // k_lm_solve_big: workgroups per ladder entry (main + helpers, solve_coop.h); 0 / 1: none
    uint32_t coop_epoch;   // ... the number of this launch among the workspace's solve launches (> 0, increasing)
    int coop_absent;       // diagnostic (MIR_LSQ_VARIANT_DEBUG_HELPERS_ABSENT): launch the main workgroups only
    int coop_rescue;       // k_lm_solve_big's RESCUE launch (one workgroup per entry, right behind the launch with helpers): an
                           // entry whose record says kQpCoopTimeout is solved again from its inputs, alone; the others return
};

// standalone BOXCQP (mir_solve_box_qp_gpu_*)
template <typename T>
struct BoxQpArgs {
    const T* P; const T* q; const T* l; const T* u; T* x;
    T relTol, absTol; uint32_t maxIterations; int unconstrained;
    SolveScratch<T> sc; int n; int f_in_lds; int* out;   // out[0] = status, out[1] = iterations
};

}  // namespace mirlsq
