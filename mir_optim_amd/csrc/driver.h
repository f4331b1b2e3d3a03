// driver.h -- host side of the MI355X-native Levenberg-Marquardt solver: what the translation units of the driver share.
//
// The control flow mirrors optimizeLeastSquaresImplGeneric!T (/root/reference/source/mir/optim/least_squares.d:877-1176,
// cited as LS:nnn) pass for pass -- validation order, Jacobian ageing, Broyden / full refresh, gradient test, damping,
// BOXCQP solve, step guard, trial acceptance, lambda / mu schedule, convergence tests -- but every floating-point
// operation of the loop runs in a HIP kernel. The host only sequences kernels on integer / boolean control state that it
// mirrors from a small device-resident LmState after each decision point.
//
//   abi.hip              the reference's extern(C) symbols (LS:637-799, boxcqp.d:36-50) + the additive entry points
//   workspace.hip        device buffers of one (m, n, element type) problem
//   solver_loop.hip      Solver<T>::run(): the LM loop, rounds, decisions (LS:930-1176)
//   solver_jacobian.hip  Jacobian refresh: finite differences (device / host callbacks, LS:1018-1049), analytic g,
//                        Broyden passes, J^T J / J^T y (LS:999-1065)
//   launch_*.hip         the translation units that instantiate the kernels (jtj_plan.h, broyden_launch.h, solve_launch.h)
//   batched.hip          one-wavefront-per-problem batched fits;  comm.hip  row-shard communicators;  unit_entries.hip
//
// There is NO CPU fallback: without a usable HIP device the solve entry points print a diagnostic and return
// status = numericError.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mir_optim_amd.h"
#include "broyden_launch.h"
#include "comm.h"
#include "common.h"
#include "jtj_plan.h"
#include "solve_launch.h"
#include "solve_types.h"

struct mir_lsq_workspace {
    size_t m = 0, n = 0, elem = 0;
    void* dev = nullptr;       // one device allocation, carved below
    size_t dev_bytes = 0;
    void* ypanel = nullptr;    // lazily allocated FD panel (device mode)
    size_t ypanel_bytes = 0;
    void* ytrial = nullptr;    // kChainMax x m trial residuals (speculative lambda ladder)
    void* ulr = nullptr;       // kLrMax x m pending Broyden columns (broyden_lr.h)
    int device = 0;            // the device the workspace lives on (callbacks' worker threads select it)
    std::vector<hipEvent_t> event_pool;   // MIR_LSQ_TIME_KERNELS: events are created once and reused by later solves
    void* pinned = nullptr;    // small pinned host block (state + trial readback), device-mapped and coherent:
    void* pinned_dev = nullptr;   // ... its device address (the decision kernels write the state mirror directly)
    void* pinned_y = nullptr;  // m-vector staging (host-callback mode), lazily allocated
    void* pinned_J = nullptr;  // m*n staging for host analytic Jacobians, lazily allocated
    // host-callback finite differences (fd_host): the 2n residual vectors of a refresh are written by the caller's f straight
    // into this pinned, point-major panel and copied to the device panel by the copy streams while other columns are
    // still being evaluated; lazily allocated
    void* pinned_panel = nullptr;
    size_t pinned_panel_bytes = 0;
    static constexpr int kCopyStreams = 4;
    hipStream_t copy_stream[kCopyStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t copy_event[kCopyStreams] = {nullptr, nullptr, nullptr, nullptr};
    int num_cu = 256;
    uint32_t solve_epoch = 0;  // solve launches with helper workgroups so far (solve_coop.h: the sync words carry it)
};

namespace mirlsq {

template <typename T> struct Abi;
template <> struct Abi<double> {
    using Settings = mir_least_squares_settings_d;
    using Result = mir_least_squares_result_d;
    using F = mir_least_squares_function_d;
    using G = mir_least_squares_jacobian_d;
    using FB = mir_lsq_batched_function_d;
};
template <> struct Abi<float> {
    using Settings = mir_least_squares_settings_s;
    using Result = mir_least_squares_result_s;
    using F = mir_least_squares_function_s;
    using G = mir_least_squares_jacobian_s;
    using FB = mir_lsq_batched_function_s;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// polite busy-wait step: the architecture's spin hint where there is one
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}

bool device_available();       // prints the "no CPU fallback" diagnostic when there is none (workspace.hip)
int query_num_cu();

// ------------------------------------------------------------------------------------------
// device buffers carved from one allocation
// ------------------------------------------------------------------------------------------
template <typename T>
struct Buffers {
    T *J, *y, *mB, *ytmp;
    T *X, *twh;
    T *x, *lower, *upper, *dx, *dx_acc, *trial, *Jy, *JJ, *packed, *partials, *sum;
    LmState<T>* st;
    ChainRec<T>* rec;
    T* slabs;
    T *lrD, *lrvec, *lrpart;   // pending Broyden steps (kLrMax x n), reduced sweep vector, per-workgroup partials
    SolveScratch<T> sc[kChainMax];
    size_t bytes;
};

constexpr int kPartials = 1024;

template <typename T> Buffers<T> carve(void* base, size_t m, size_t n, int num_cu);
template <typename T> mir_lsq_workspace* workspace_create(size_t m, size_t n);
void workspace_destroy(mir_lsq_workspace* ws);

// ------------------------------------------------------------------------------------------
// the solver
// ------------------------------------------------------------------------------------------
struct EventPair { hipEvent_t a, b; int kind; };   // kind: 0 jtj, 1 Broyden pass, 2 solve, 3 FD jtj, 4 FD callbacks, 5 trial callbacks

template <typename T>
struct Solver {
    using Settings = typename Abi<T>::Settings;
    using Result = typename Abi<T>::Result;
    using F = typename Abi<T>::F;
    using G = typename Abi<T>::G;
    using FB = typename Abi<T>::FB;

    const Settings* S;
    size_t m;
    uint32_t n;
    T* xh;                 // caller's x (host), updated in place on accepted steps (LS:1135)
    const T *lh, *uh;
    void* fctx; F f;
    void* gctx; G g;
    void* tmctx; mir_least_squares_thread_manager tm;
    void* fbctx; FB fb;
    FB fbr = nullptr;          // batched residual callback writing Y row-major (m x p): the pair panel of the fused FD kernels
    FB fbd = nullptr;          // batched residual callback writing the m x n row-major DIFFERENCE panel (fbRowMajorDiff)
    int fd_fused = 0;          // 1: the (+h, -h) pair panel, 2: the difference panel of this refresh is in ws->ypanel and J has not been filled yet
    int sums_pending = 0;      // > 0: the trial sums of this round are still k_lr_sumsq's stage-1 partials (k_decide_chain finishes them)
    uint32_t fd_batch;
    bool device_cb;
    bool time_kernels;
    mir_lsq_comm* comm;
    mir_lsq_stats* stats;
    mir_lsq_trace* trace = nullptr;

    hipStream_t stream = nullptr;
    bool own_stream = false;
    mir_lsq_workspace* ws = nullptr;
    bool own_ws = false;
    Buffers<T> B;
    JtjPlan plan;
    LmSettingsDev<T> sd;
    // A/B switches (mir_lsq_gpu_options.variant, MIR_LSQ_VARIANT_*): 0 = product path. Broyden passes keep J and carry the
    // updates as pending rank-one terms (broyden_lr.h); BROYDEN_REWRITE selects the kernels that rewrite J every pass;
    // bits 16..20 the number of pending terms after which they are folded into J
    uint32_t variant = 0;
    bool dbg_solve = false, no_speculation = false, lowrank = true, no_null_skip = false, host_profile = false;
    mir_lsq_stats stats_local{};      // the solve works on this image; stats_bytes of it go back to the caller's struct
    mir_lsq_stats* stats_user = nullptr;
    size_t stats_bytes = 0;
    uint64_t launches_mark = 0;       // tl_launches at the start of the round being accounted (mir_lsq_stats.round_launches)
    int round_kind = -1;
    uint64_t launches_excluded = 0;   // of the launches since the mark: those that belong to no round (flush + resynchronisation)
    std::atomic<uint64_t> worker_launches{0};   // kernels launched on behalf of this solve by OTHER threads (the thread manager's
                                                // workers, FD_HOST_COLUMNS): tl_launches is per thread and does not see them
    int lr_cap = kLrMax;
    int lr_k = 0;
    int device = 0, caller_device = -1;
    // the reference swaps the contents of y and mBuffer on acceptance (LS:1136); here the two device buffers swap roles.
    // Everything that touches them goes through these members (never through B.y / B.mB directly).
    T* y = nullptr;
    T* mB = nullptr;
    T* fr = nullptr;     // third m-vector: the trial residual of a round goes here; accepting rotates (y, mB, fr) <- (fr, y, mB)
    // null steps (trial == x bit for bit; kFlagNullStep): once a round ended on one, the next round's solves are looked
    // at before the callbacks are launched, and when every entry is a null step nothing is evaluated
    bool has_bounds = true;    // some lower / upper entry is finite (set in run()); MIR_LSQ_VARIANT_SOLVE_BOUNDED forces the full kernel
    bool tail_null = false;
    // ---- the fused round (solver_loop.hip, setup() and enqueue_fused_tail()): behind a round's one trial residual the next
    // pass's Broyden sweep runs speculatively and carries the trial's sum of squares; one kernel then decides the trial and, if
    // it is accepted without any exit test firing, applies the pass's n x n side and solves the next system. Results are
    // bit-identical with the one-by-one rounds (MIR_LSQ_VARIANT_NO_PIPELINE; tests/test_gpu_lm.py).
    bool fused = false;            // allowed at all for this solve (see setup())
    bool spec_enqueue = false;     // set while the kernels of the pass run ahead are being enqueued (their events are tagged)
    size_t spec_events_from = 0;   // events of the pass run ahead start here
    bool big_solve = false;    // n > 256 (or MIR_LSQ_VARIANT_SOLVE_GENERIC): the any-n solve kernel
    bool coop_disabled = false;   // helper workgroups timed out once in this solve: one workgroup per entry from now on

    LmState<T>* st_h;      // pinned mirror of the decision point being processed (one of st_slot[])
    LmState<T>* st_slot[2] = {nullptr, nullptr};     // the two mirrors, host and device addresses
    LmState<T>* st_slot_d[2] = {nullptr, nullptr};
    T* x_slot[2] = {nullptr, nullptr};               // accepted point of decision point seq at x_slot[seq & 1]
    T* x_slot_d[2] = {nullptr, nullptr};
    T* x_h = nullptr;              // x mirror of the decision point being processed
    uint32_t seq = 0;              // decision points enqueued so far
    T* trial_h;            // pinned, n
    std::vector<T> twh_h;
    std::vector<EventPair> events;
    Result ret;

    // host-callback finite differences (reference thread-manager contract, LS:1019-1048)
    struct Slot { T* p = nullptr; T* yp = nullptr; T* ym = nullptr; };
    std::vector<Slot> slots;
    std::vector<int> slot_count;
    std::mutex fd_mutex;
    std::atomic<bool> fd_failed{false};
    bool fd_panel_mode = false;        // this refresh stages through the pinned point-major panel (fd_host)
    std::atomic<uint32_t> fd_streams_used{0};   // bit k: copy stream k carries copies of this refresh
    std::atomic<uint64_t> fd_f_ns{0};  // wall time inside the caller's f, summed over the tasks (statistics)
    std::atomic<uint32_t> fd_tasks_run{0};   // tasks the manager has run in this refresh: every i in [0, n) exactly once (LS:575-578)

    // MIR_LSQ_VARIANT_HOST_PROFILE: host wall time per category of runtime call, printed at teardown (diagnostic)
    double hp_ms[6] = {0, 0, 0, 0, 0, 0};   // 0 events, 1 all-reduce calls, 2 callbacks, 3 sync/readback, 4 launches (solve), 5 max single
    struct HpScope {
        Solver* s; int cat; std::chrono::steady_clock::time_point t0;
        HpScope(Solver* s_, int c) : s(s_), cat(c) { if (s->host_profile) t0 = std::chrono::steady_clock::now(); }
        ~HpScope() {
            if (!s->host_profile) return;
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            s->hp_ms[cat] += ms;
            if (ms > s->hp_ms[5]) s->hp_ms[5] = ms;
        }
    };

    bool ok(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return true;
        std::fprintf(stderr, "[mir_optim_amd] %s failed: %s\n", what, hipGetErrorName(e));
        return false;
    }
    // kernels launched for this solve so far: this thread's counter + what the thread manager's workers launched
    uint64_t launches_now() const { return tl_launches + worker_launches.load(std::memory_order_relaxed); }
    void close_round()
    {
        const uint64_t now = launches_now();
        if (stats) {
            stats->library_launches += now - launches_mark;
            if (round_kind >= 0) { stats->round_launches[round_kind] += now - launches_mark - launches_excluded; stats->rounds[round_kind]++; }
        }
        launches_mark = now;
        launches_excluded = 0;
        round_kind = -1;
    }
    void ev_begin(int kind);
    void ev_end();

    // ---- solver_loop.hip
    bool setup();
    void teardown();
    bool eval_f(const T* x_dev, const T* x_host, T* y_dev);
    int sumsq_blocks() const;
    bool sumsq(const T* v, int slot);
    bool trial_sums(const T* v, int count, size_t vstride);
    bool allreduce(T* buf, size_t count, int kind);
    void trace_emit(int event, uint32_t iterations, T lambda, T residual, T trial_residual, T dx_dot);
    bool trace_round(int ks, T residual_before, uint32_t iterations_before);
    bool wait_state(uint32_t expect);
    bool read_state(const T* vec_dev);
    void print_solve_dbg(bool fused_round);
    LmSolveArgs<T> solve_args(int ks, const T* lam, bool check_grad, bool lambda_from_state);
    bool enqueue_solve(int ks, const T* lam, bool check_grad, bool lambda_from_state);
    DecideArgs<T> decide_args(int ks, bool check_grad, bool lambda_from_state);
    bool enqueue_decide(int ks, bool check_grad, bool lambda_from_state, const T* sum_v = nullptr);
    bool enqueue_fused_tail(const T* ytr, bool check_grad, bool lambda_from_state);
    void commit_spec_round();
    void drop_spec_round();
    Result run();

    // ---- solver_jacobian.hip
    bool broyden_lowrank(const T* y_dev, const T* yold_dev);
    bool plain_products(const T* y_vec);
    bool unpack_in_reduce(bool fd, bool broyden = false) const { return !comm && (fd || jtj_plain_unpacks(plan, broyden)); }
    JtjUnpack<T> unpack_target() { JtjUnpack<T> u; u.JJ = B.JJ; u.Jy = B.Jy; return u; }
    bool finish_products(bool direct);
    bool jacobian_products(bool broyden, const T* y_dev, const T* yold_dev);
    bool fd_device();
    static void fd_task_trampoline(mir_least_squares_task task, uint32_t totalThreads, uint32_t threadId, uint32_t j);
    void fd_task(uint32_t totalThreads, uint32_t threadId, uint32_t j);
    bool fd_host_prepare_panel();
    bool fd_host();
    bool analytic_jacobian();
};

// the one entry every solve goes through (solver_loop.hip; instantiated for double and float)
template <typename T>
typename Abi<T>::Result solve_entry(const typename Abi<T>::Settings* settings, size_t m, size_t n, T* x, const T* l,
                                    const T* u, const mir_lsq_gpu_options* opt, void* fctx, typename Abi<T>::F f,
                                    void* gctx, typename Abi<T>::G g, void* tmctx, mir_least_squares_thread_manager tm);

}  // namespace mirlsq
