// lm_driver.hip -- host side of the MI355X-native Levenberg-Marquardt solver and its C ABI.
//
// The control flow mirrors optimizeLeastSquaresImplGeneric!T
// (/root/reference/source/mir/optim/least_squares.d:877-1176, cited as LS:nnn below) pass for
// pass -- validation order, Jacobian ageing, Broyden/full refresh, gradient test, damping,
// BOXCQP solve, step guard, trial acceptance, lambda/mu schedule, convergence tests -- but every
// floating-point operation of the loop runs in a HIP kernel (jtj_kernel.h, solve_kernel.h,
// misc_kernels.h). The host only sequences kernels on integer/boolean control state that it
// mirrors from a small device-resident LmState after each decision point.
//
// There is NO CPU fallback: without a usable HIP device the solve entry points print a
// diagnostic and return status = numericError.
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
#include "comm.h"
#include "common.h"
#include "jtj_launch.h"
#include "batched_kernel.h"
#include "broyden_lr.h"
#include "misc_kernels.h"
#include "solve_kernel.h"
#include "solve_big.h"

using namespace mirlsq;

// ------------------------------------------------------------------------------------------
// ABI pins
// ------------------------------------------------------------------------------------------
static_assert(sizeof(mir_least_squares_settings_d) == 128, "Settings_d must be 128 bytes");
static_assert(offsetof(mir_least_squares_settings_d, jacobianEpsilon) == 8, "");
static_assert(offsetof(mir_least_squares_settings_d, lambdaDecrease) == 96, "");
static_assert(offsetof(mir_least_squares_settings_d, qpSettings) == 104, "");
static_assert(sizeof(mir_least_squares_settings_s) == 68, "Settings_s must be 68 bytes");
static_assert(offsetof(mir_least_squares_settings_s, qpSettings) == 56, "");
static_assert(sizeof(mir_least_squares_result_d) == 32, "Result_d must be 32 bytes");
static_assert(offsetof(mir_least_squares_result_d, residual) == 16, "");
static_assert(sizeof(mir_least_squares_result_s) == 24, "Result_s must be 24 bytes");
static_assert(sizeof(mir_slice_d) == 16 && sizeof(mir_least_squares_task) == 16, "");

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
    // two-stream finite-difference refresh (fbRowMajorDiffWindow): the side stream the caller's window kernels run on, one
    // event per window (+ one that releases the side stream), per-window slab sets; all lazily created
    static constexpr int kMaxWindows = 16;
    hipStream_t side_stream = nullptr;
    hipEvent_t win_event[kMaxWindows + 1] = {};
    void* win_slabs = nullptr;
    size_t win_slab_bytes = 0;
    static constexpr int kCopyStreams = 4;
    hipStream_t copy_stream[kCopyStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t copy_event[kCopyStreams] = {nullptr, nullptr, nullptr, nullptr};
    int num_cu = 256;
};

namespace {

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

bool device_available()
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        std::fprintf(stderr, "[mir_optim_amd] no usable HIP device: the MI355X kernels cannot run "
                             "(this library has no CPU fallback)\n");
        return false;
    }
    return true;
}

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
    T* lrranges;               // kReduceRanges x lr_len(n): range sums of the sweep's fused reduction (broyden_lr.h, lr_tail)
    uint32_t* counters;        // kCounters arrival counters of the "last workgroup finishes" tails; zero between launches
    SolveScratch<T> sc[kChainMax];
    size_t bytes;
};

constexpr int kPartials = 1024;
// arrival counters: [0, kReduceRanges] the Broyden sweep's ranges + top, then the one of the sum of squares
constexpr int kCounters = 64, kCounterSumsq = 40;
static_assert(kReduceRanges + 1 <= kCounterSumsq, "counter block layout");

// slab elements the J^T J kernels may need for this shape: the larger of the product plan and the streaming variant's
template <typename T>
size_t slab_elems(size_t m, size_t n, int num_cu)
{
    const JtjPlan a = jtj_plan<T>(m, (int)n, num_cu, 0), b = jtj_plan<T>(m, (int)n, num_cu, MIR_LSQ_VARIANT_JTJ_STREAM);
    const size_t ea = (size_t)a.nblk * a.njobs * a.slab_len, eb = (size_t)b.nblk * b.njobs * b.slab_len;
    const size_t ec = (size_t)a.fdp8_nblk * a.fdp8_slab_len;
    const size_t ed = (size_t)a.pc32_nblk * a.slab_len;
    size_t e = ea > eb ? ea : eb;
    e = e > ec ? e : ec;
    return e > ed ? e : ed;
}

template <typename T>
Buffers<T> carve(void* base, size_t m, size_t n, int num_cu)
{
    Buffers<T> b{};
    long long* dbg_all = nullptr;
    size_t off = 0;
    auto take = [&](size_t count, size_t elem) {
        void* p = base ? static_cast<char*>(base) + off : nullptr;
        off = align_up(off + count * elem, 256);
        return p;
    };
    b.J = (T*)take(m * n, sizeof(T));
    b.y = (T*)take(m, sizeof(T));
    b.mB = (T*)take(m, sizeof(T));
    b.ytmp = (T*)take(m, sizeof(T));
    b.X = (T*)take(2 * n * n, sizeof(T));
    b.twh = (T*)take(n, sizeof(T));
    b.x = (T*)take(n, sizeof(T));
    b.lower = (T*)take(n, sizeof(T));
    b.upper = (T*)take(n, sizeof(T));
    b.dx = (T*)take(kChainMax * n, sizeof(T));
    b.dx_acc = (T*)take(n, sizeof(T));
    b.trial = (T*)take(kChainMax * n, sizeof(T));
    b.Jy = (T*)take(n, sizeof(T));
    b.JJ = (T*)take(n * n, sizeof(T));
    b.packed = (T*)take(n * (n + 1) / 2 + n + 8, sizeof(T));
    b.partials = (T*)take((size_t)kChainMax * kPartials, sizeof(T));
    b.sum = (T*)take(8 + kChainMax, sizeof(T));
    b.st = (LmState<T>*)take(1, sizeof(LmState<T>));
    b.rec = (ChainRec<T>*)take(kChainMax, sizeof(ChainRec<T>));
    b.slabs = (T*)take(slab_elems<T>(m, n, num_cu), sizeof(T));
    b.lrD = (T*)take((size_t)kLrMax * n, sizeof(T));
    b.lrvec = (T*)take((size_t)lr_len((int)n) + 6, sizeof(T));
    b.lrpart = (T*)take((size_t)lr_blocks(m, num_cu) * lr_len((int)n), sizeof(T));
    b.lrranges = (T*)take((size_t)kReduceRanges * lr_len((int)n), sizeof(T));
    b.counters = (uint32_t*)take(kCounters, sizeof(uint32_t));
    for (int k = 0; k < kChainMax; ++k) {
        b.sc[k].Pm = (T*)take(n * n, sizeof(T));
        b.sc[k].A = (T*)take(n * n, sizeof(T));
        b.sc[k].Fg = (T*)take(n * (n | 1), sizeof(T));
        b.sc[k].vec = (T*)take(12 * n, sizeof(T));
        b.sc[k].ivec = (int32_t*)take(2 * n, sizeof(int32_t));
        b.sc[k].dbg = nullptr;
    }
    b.sc[0].dbg = (long long*)take(32, sizeof(long long));
    dbg_all = b.sc[0].dbg;
    (void)dbg_all;
    b.bytes = off;
    return b;
}

int query_num_cu()
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

void workspace_destroy(mir_lsq_workspace* ws);

template <typename T>
mir_lsq_workspace* workspace_create(size_t m, size_t n)
{
    auto* ws = new mir_lsq_workspace();
    ws->m = m; ws->n = n; ws->elem = sizeof(T);
    ws->num_cu = query_num_cu();
    const Buffers<T> sz = carve<T>(nullptr, m, n, ws->num_cu);
    ws->dev_bytes = sz.bytes;
    if (hipMalloc(&ws->dev, ws->dev_bytes) != hipSuccess) {
        std::fprintf(stderr, "[mir_optim_amd] hipMalloc(%zu bytes) failed\n", ws->dev_bytes);
        delete ws;
        return nullptr;
    }
    // the m-sized side buffers of the solve loop are part of the workspace (no allocation inside a solve): the pending
    // Broyden columns (kLrMax x m) and the trial residuals of the lambda ladder (kChainMax x m)
    if (hipGetDevice(&ws->device) != hipSuccess) ws->device = 0;
    if (hipMalloc(&ws->ulr, (size_t)kLrMax * m * sizeof(T)) != hipSuccess
        || hipMalloc(&ws->ytrial, (size_t)kChainMax * m * sizeof(T)) != hipSuccess
        || hipHostMalloc(&ws->pinned, 2 * sizeof(LmState<T>) + (3 * n + 8) * sizeof(T) + 3 * align_up(n * sizeof(T), 256) + 256, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess
        || hipHostGetDevicePointer(&ws->pinned_dev, ws->pinned, 0) != hipSuccess) {
        std::fprintf(stderr, "[mir_optim_amd] workspace side buffers: allocation failed\n");
        workspace_destroy(ws);
        return nullptr;
    }
    return ws;
}

void workspace_destroy(mir_lsq_workspace* ws)
{
    if (!ws) return;
    for (hipEvent_t e : ws->event_pool) (void)hipEventDestroy(e);
    if (ws->dev) (void)hipFree(ws->dev);
    if (ws->ypanel) (void)hipFree(ws->ypanel);
    if (ws->ytrial) (void)hipFree(ws->ytrial);
    if (ws->ulr) (void)hipFree(ws->ulr);
    if (ws->pinned) (void)hipHostFree(ws->pinned);
    if (ws->pinned_y) (void)hipHostFree(ws->pinned_y);
    if (ws->pinned_J) (void)hipHostFree(ws->pinned_J);
    if (ws->pinned_panel) (void)hipHostFree(ws->pinned_panel);
    if (ws->win_slabs) (void)hipFree(ws->win_slabs);
    for (hipEvent_t e : ws->win_event) if (e) (void)hipEventDestroy(e);
    if (ws->side_stream) (void)hipStreamDestroy(ws->side_stream);
    for (int k = 0; k < mir_lsq_workspace::kCopyStreams; ++k) {
        if (ws->copy_event[k]) (void)hipEventDestroy(ws->copy_event[k]);
        if (ws->copy_stream[k]) (void)hipStreamDestroy(ws->copy_stream[k]);
    }
    delete ws;
}

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
    FB fbr = nullptr;          // batched residual callback writing Y row-major (m x p): finite differences fused into k_jtj2
    FB fbd = nullptr;          // batched residual callback writing the m x n row-major DIFFERENCE panel (fbRowMajorDiff)
    mir_lsq_window_function_d fbdw = nullptr;   // ... its row-window form, for the two-stream refresh (f64)
    uint32_t fd_windows = 0;
    int fd_fused = 0;          // 1: the (+h, -h) pair panel, 2: the difference panel of this refresh is in ws->ypanel and J has not been filled yet
    int sums_pending = 0;      // > 0: the trial sums of this round are still stage-1 partials (k_decide_chain finishes them)
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
    // updates as pending rank-one terms (broyden_lr.h); BROYDEN_REWRITE selects the kernels that rewrite J every pass
    // (k_jtj2<., true> / k_jtj8 / k_broyden_wide); bits 16..20 the number of pending terms after which they are folded into J
    uint32_t variant = 0;
    bool dbg_solve = false, no_speculation = false, lowrank = true, no_null_skip = false, host_profile = false;
    // Fewer launches per pass. merge_small (default; MIR_LSQ_VARIANT_NO_TAIL_FUSION = round 2's sequence):
    // the slab reduction writes J^T J / J^T y itself and the solve kernel takes |J^T y|_inf in its prologue (and, opt-in
    // MIR_LSQ_VARIANT_FINISH_IN_SOLVE, applies the n x n finish of a Broyden pass there). sweep_tail / sumsq_tail (opt-in,
    // MIR_LSQ_VARIANT_SWEEP_TAIL / SUMSQ_TAIL): "last workgroup finishes" tails inside the Broyden sweep and the sum-of-squares sweep.
    bool merge_small = true, sweep_tail = false, sumsq_tail = false;
    // The Jacobian the Broyden sweeps read: B.J, or -- MIR_LSQ_VARIANT_FD_PANEL_IS_J, after a refresh through the difference
    // panel -- the PANEL ITSELF with the column widths twh (broyden_lr.h, LrArgs::colscale): the fused finite-difference kernel
    // then does not write J at all (1 GB less HBM traffic per refresh at cfg 3); J is materialised only when the pending
    // terms are folded into it.
    const T* Jcur = nullptr;
    const T* Jscale = nullptr;
    bool finish_pending = false;   // a Broyden sweep's reduced vector waits in B.lrvec for the solve kernel's prologue
    int finish_k = 0;              // ... with this many pending terms before it
    mir_lsq_stats stats_local{};      // the solve works on this image; stats_bytes of it go back to the caller's struct
    mir_lsq_stats* stats_user = nullptr;
    size_t stats_bytes = 0;
    uint64_t launches_mark = 0;       // tl_launches at the start of the round being accounted (mir_lsq_stats.round_launches)
    int round_kind = -1;
    uint64_t launches_excluded = 0;   // of the launches since the mark: those that belong to no round (flush + resynchronisation)
    void close_round()
    {
        const uint64_t now = tl_launches;
        if (stats) {
            stats->library_launches += now - launches_mark;
            if (round_kind >= 0) { stats->round_launches[round_kind] += now - launches_mark - launches_excluded; stats->rounds[round_kind]++; }
        }
        launches_mark = now;
        launches_excluded = 0;
        round_kind = -1;
    }
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
    bool has_bounds = true;    // some lower / upper entry is finite (set in run()); MIR_LSQ_SOLVE_BOUNDED=1 forces the full kernel
    bool tail_null = false;
    // ---- rounds enqueued ahead of time (pipelining of the host): while the GPU runs round r, the host already enqueues
    // the round that follows IF r is accepted without any exit test firing -- Broyden sweep, solve, trial residual,
    // decision -- behind a device-side guard (LmState::spec_ok, set by k_decide_chain of round r). If r ends differently the
    // guarded kernels return at once and the host enqueues the right round as before. Results are bit-identical with and
    // without it (tests/test_gpu_lm.py); what disappears is the launch latency between accepted rounds.
    bool spec_enqueue = false;     // set while the kernels of such a round are being enqueued
    bool pipeline = true;          // allowed at all for this solve (see setup())
    size_t spec_events_from = 0;   // events of the round enqueued ahead of time start here
    int f_in_lds = 0;
    bool big_solve = false;    // n > 256 (or MIR_LSQ_VARIANT_SOLVE_GENERIC): the any-n solve kernel
    int solve_nb_ = 0;
    size_t solve_lds = 0;

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

    // MIR_LSQ_HOST_PROFILE=1: host wall time per category of runtime call, printed at teardown (diagnostic)
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

    void ev_begin(int kind)
    {
        if (!time_kernels) return;
        HpScope hp(this, 0);
        EventPair p{};
        p.kind = spec_enqueue ? kind + 100 : kind;       // rounds enqueued ahead of time: counted only once they are committed
        // events come from the workspace's pool (created on first use, reused by every later solve on this workspace)
        if (ws->event_pool.size() < 2 * (events.size() + 1)) {
            hipEvent_t ea, eb;
            (void)hipEventCreate(&ea);
            (void)hipEventCreate(&eb);
            ws->event_pool.push_back(ea);
            ws->event_pool.push_back(eb);
        }
        p.a = ws->event_pool[2 * events.size()];
        p.b = ws->event_pool[2 * events.size() + 1];
        (void)hipEventRecord(p.a, stream);
        events.push_back(p);
    }
    void ev_end()
    {
        if (!time_kernels) return;
        HpScope hp(this, 0);
        (void)hipEventRecord(events.back().b, stream);
    }

    bool setup()
    {
        if (!ws) {
            ws = workspace_create<T>(m, n);
            own_ws = true;
            if (!ws) return false;
        } else if (ws->m != m || ws->n != n || ws->elem != sizeof(T)) {
            std::fprintf(stderr, "[mir_optim_amd] workspace shape mismatch\n");
            ws = nullptr;
            return false;
        }
        // the workspace's device becomes current for the call (restored at teardown): kernels, streams and the dynamic-LDS
        // attributes are per device, and a caller may drive several devices from one process
        if (hipGetDevice(&caller_device) != hipSuccess) caller_device = ws->device;
        if (caller_device != ws->device && !ok(hipSetDevice(ws->device), "hipSetDevice")) return false;
        plan = jtj_plan<T>(m, (int)n, ws->num_cu, variant);
        B = carve<T>(ws->dev, m, n, ws->num_cu);
        device = ws->device;
        // pinned block: [state mirror 0 | state mirror 1 | x mirror 0 | x mirror 1 | staging vector]; decision point `seq`
        // is published into mirror seq & 1, so a round enqueued ahead of time cannot overwrite the one the host still reads
        for (int k = 0; k < 2; ++k) {
            st_slot[k] = reinterpret_cast<LmState<T>*>(ws->pinned) + k;
            st_slot_d[k] = reinterpret_cast<LmState<T>*>(ws->pinned_dev) + k;
            x_slot[k] = reinterpret_cast<T*>(static_cast<char*>(ws->pinned) + 2 * sizeof(LmState<T>)) + (size_t)k * n;
            x_slot_d[k] = reinterpret_cast<T*>(static_cast<char*>(ws->pinned_dev) + 2 * sizeof(LmState<T>)) + (size_t)k * n;
            st_slot[k]->seq = 0;
        }
        st_h = st_slot[0];
        trial_h = reinterpret_cast<T*>(static_cast<char*>(ws->pinned) + 2 * sizeof(LmState<T>)) + 2 * (size_t)n;   // staging vector
        if (!stream) {
            if (!ok(hipStreamCreate(&stream), "hipStreamCreate")) return false;
            own_stream = true;
        }
        sd.jacobianEpsilon = S->jacobianEpsilon; sd.absTolerance = S->absTolerance; sd.relTolerance = S->relTolerance;
        sd.gradTolerance = S->gradTolerance; sd.maxGoodResidual = S->maxGoodResidual; sd.maxStep = S->maxStep;
        sd.maxLambda = S->maxLambda; sd.minLambda = S->minLambda; sd.minStepQuality = S->minStepQuality;
        sd.goodStepQuality = S->goodStepQuality; sd.lambdaIncrease = S->lambdaIncrease; sd.lambdaDecrease = S->lambdaDecrease;
        sd.qpRelTolerance = S->qpSettings.relTolerance; sd.qpAbsTolerance = S->qpSettings.absTolerance;
        sd.qpMaxIterations = S->qpSettings.maxIterations; sd.pad = 0;
        dbg_solve = (variant & MIR_LSQ_VARIANT_DEBUG_SOLVE) != 0;
        no_speculation = (variant & MIR_LSQ_VARIANT_NO_SPECULATION) != 0;
        lowrank = (variant & MIR_LSQ_VARIANT_BROYDEN_REWRITE) == 0;
        no_null_skip = (variant & MIR_LSQ_VARIANT_NO_NULL_SKIP) != 0;
        merge_small = (variant & MIR_LSQ_VARIANT_NO_TAIL_FUSION) == 0;
        sweep_tail = merge_small && (variant & MIR_LSQ_VARIANT_SWEEP_TAIL) != 0;
        sumsq_tail = merge_small && (variant & MIR_LSQ_VARIANT_SUMSQ_TAIL) != 0;
        finish_pending = false;
        host_profile = (variant & MIR_LSQ_VARIANT_HOST_PROFILE) != 0;
        {
            const int v = (int)((variant >> MIR_LSQ_VARIANT_LR_CAP_SHIFT) & 31u);
            if (v >= 1 && v <= kLrMax) lr_cap = v;
        }
        y = B.y;
        mB = B.mB;
        fr = B.ytmp;
        Jcur = B.J;
        Jscale = nullptr;
        big_solve = n > (uint32_t)kSolveMaxN || (variant & MIR_LSQ_VARIANT_SOLVE_GENERIC) != 0;
        if (n > (uint32_t)kLrMaxN) lowrank = false;    // the read-only Broyden sweep keeps n <= 256; above, J is rewritten
        // opt-in (MIR_LSQ_VARIANT_PIPELINE): measured on one MI355X it does not pay -- cfg 3 7.08 ms per solve with it, 7.02
        // without; cfg 2 2.92 against 3.00 ms (scripts/ab_bench.sh) -- the stream is already busy > 97 % of a solve
        // ... except for SMALL problems (J up to 32 MB: every kernel of a round is a few microseconds and the host's decision
        // latency is a visible share of it): cfg 2 63.5 -> 61.8 us per round, so those pipeline by default
        // (MIR_LSQ_VARIANT_NO_PIPELINE turns it off); at a strong-scaled rank's 125 000 x 128 it changes nothing (2.72 ms either way)
        const bool small_problem = (double)m * (double)n * sizeof(T) <= 32.0 * 1024 * 1024;
        pipeline = device_cb && lowrank && !big_solve && !trace && !dbg_solve
            && ((variant & MIR_LSQ_VARIANT_PIPELINE) || (small_problem && !(variant & MIR_LSQ_VARIANT_NO_PIPELINE)))
            && (!comm || comm->kind == 1);      // host-mediated communicators synchronise the stream inside every exchange
        solve_nb_ = solve_nb((int)n, (int)sizeof(T));
        f_in_lds = solve_nb_ > 0;
        solve_lds = solve_lds_bytes((int)n, (int)sizeof(T));
        twh_h.resize(n);
        // arrival counters of the opt-in tails: every tail leaves its counter at zero, but a solve that died half way may not have
        if ((sweep_tail || sumsq_tail) && !ok(hipMemsetAsync(B.counters, 0, kCounters * sizeof(uint32_t), stream), "memset counters")) return false;
        // x, lower, upper sit back to back in the workspace: one copy from the pinned block instead of three from pageable
        // memory (each of those is a staged blit kernel, ~18 us apart on the stream)
        {
            char* stage = static_cast<char*>(ws->pinned) + align_up(2 * sizeof(LmState<T>) + (3 * (size_t)n + 8) * sizeof(T), 256);
            const size_t ol = (size_t)(reinterpret_cast<char*>(B.lower) - reinterpret_cast<char*>(B.x));
            const size_t ou = (size_t)(reinterpret_cast<char*>(B.upper) - reinterpret_cast<char*>(B.x));
            std::memcpy(stage, xh, n * sizeof(T));
            std::memcpy(stage + ol, lh, n * sizeof(T));
            std::memcpy(stage + ou, uh, n * sizeof(T));
            return ok(hipMemcpyAsync(B.x, stage, ou + n * sizeof(T), hipMemcpyHostToDevice, stream), "H2D x, l, u");
        }
    }

    void teardown()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        if (host_profile) std::fprintf(stderr, "[host profile] J %p  FD panel %p  y %p\n", (void*)B.J, ws->ypanel, (void*)B.y);
        if (host_profile)
            std::fprintf(stderr, "[host profile] events %.3f ms  all-reduce calls %.3f  trial callbacks %.3f  readback+sync %.3f  solve launch %.3f  "
                                 "longest single call %.3f\n", hp_ms[0], hp_ms[1], hp_ms[2], hp_ms[3], hp_ms[4], hp_ms[5]);
        if (stats) {
            for (auto& e : events) {
                if (e.kind < 0 || e.kind >= 100) continue;       // a round enqueued ahead of time whose guard stayed closed
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e.a, e.b);
                if (e.kind == 0) { stats->jtj_ms += ms; stats->jtj_launches++; }
                else if (e.kind == 1) { stats->jtj_ms += ms; stats->jtj_launches++; stats->jtj_broyden_ms += ms; stats->jtj_broyden_launches++; }
                else if (e.kind == 2) { stats->solve_ms += ms; stats->solve_launches++; }
                else if (e.kind == 3) { stats->jtj_ms += ms; stats->jtj_launches++; stats->jtj_fd_ms += ms; stats->jtj_fd_launches++; }
                else if (e.kind == 4) stats->fd_callback_ms += ms;          // the calls themselves are counted where they are made
                else if (e.kind == 5) stats->trial_callback_ms += ms;
            }
        }
        events.clear();                                  // the events themselves stay in the workspace's pool
        for (auto& s : slots) {
            if (s.p) std::free(s.p);
            if (s.yp) (void)hipHostFree(s.yp);
            if (s.ym) (void)hipHostFree(s.ym);
        }
        slots.clear();
        if (own_stream && stream) (void)hipStreamDestroy(stream);
        if (ws && caller_device >= 0 && caller_device != ws->device) (void)hipSetDevice(caller_device);
        if (own_ws && ws) workspace_destroy(ws);
    }

    // ---- residual evaluation f(x) -> y_dev.  x_dev / x_host describe the same point.
    bool eval_f(const T* x_dev, const T* x_host, T* y_dev)
    {
        if (device_cb) {
            f(fctx, m, n, x_dev, y_dev);
            return true;
        }
        if (!ws->pinned_y && !ok(hipHostMalloc(&ws->pinned_y, m * sizeof(T), hipHostMallocDefault), "hipHostMalloc(y)")) return false;
        T* yh = static_cast<T*>(ws->pinned_y);
        const auto t0 = std::chrono::steady_clock::now();
        f(fctx, m, n, x_host, yh);
        if (stats) { stats->host_f_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); stats->host_f_calls++; }
        return ok(hipMemcpyAsync(y_dev, yh, m * sizeof(T), hipMemcpyHostToDevice, stream), "H2D y")
            && ok(hipStreamSynchronize(stream), "sync");
    }

    template <int NB, bool BOUNDED>
    hipError_t launch_solve_nbb(const LmSolveArgs<T>& a, int ks)
    {
        auto kern = k_lm_solve<T, NB, BOUNDED>;
        if (solve_lds > 48 * 1024) MIRLSQ_ENSURE_LDS(kern, (size_t)(160 * 1024 - 256));
        MIRLSQ_LAUNCH(kern, dim3(ks), dim3(kSolveThreads), solve_lds, stream, a);
        return hipGetLastError();
    }
    template <int NB>
    hipError_t launch_solve_nb(const LmSolveArgs<T>& a, int ks)
    {
        // all bounds infinite: the variant without the BOXCQP active-set loop (solve_kernel.h, BOUNDED = false)
        return has_bounds ? launch_solve_nbb<NB, true>(a, ks) : launch_solve_nbb<NB, false>(a, ks);
    }
    hipError_t launch_solve(const LmSolveArgs<T>& a, int ks)
    {
        if (big_solve) {
            // any n: one 512-thread workgroup per ladder entry, matrices in global memory (solve_big.h)
            auto kern = k_lm_solve_big<T>;
            MIRLSQ_ENSURE_LDS(kern, sizeof(BigLds<T>));
            MIRLSQ_LAUNCH(kern, dim3(ks), dim3(kBigThreads), sizeof(BigLds<T>), stream, a);
            return hipGetLastError();
        }
        switch (solve_nb_) {
        case 1: return launch_solve_nb<1>(a, ks);
        case 2: return launch_solve_nb<2>(a, ks);
        case 4: return launch_solve_nb<4>(a, ks);
        case 8: return launch_solve_nb<8>(a, ks);
        default: return launch_solve_nb<0>(a, ks);
        }
    }

    // ---- ||v_k||^2 for k < count vectors (stride vstride) -> B.sum[slot + k] on device (all-reduced over row shards)
    //      defer_final (single GPU, trial sums): stage 2 is left to k_decide_chain (sums_pending = the partial count)
    int sumsq_blocks() const
    {
        int nb = (int)((m + 4095) / 4096);
        if (nb > kPartials) nb = kPartials;
        return nb < 1 ? 1 : nb;
    }
    SumsqTailArgs<T> sumsq_tail_args(const T* v, int slot, size_t vstride)
    {
        SumsqTailArgs<T> t{};
        t.v = v; t.m = m; t.vstride = vstride; t.partials = B.partials; t.pstride = kPartials;
        t.counter = B.counters + kCounterSumsq; t.out = B.sum + slot;
        return t;
    }
    bool sumsq(const T* v, int slot, int count = 1, size_t vstride = 0, bool defer_final = false)
    {
        const int nb = sumsq_blocks();
        if (sumsq_tail) {
            // stage 2 in the last workgroup of stage 1 (same fixed order): one launch
            MIRLSQ_LAUNCH((k_sumsq_tail<T, kSumsqTailFinal>), dim3(nb, count), dim3(kSolveThreads), 0, stream, sumsq_tail_args(v, slot, vstride));
            if (comm && !allreduce(B.sum + slot, (size_t)count, 2)) return false;
            return ok(hipGetLastError(), "sumsq");
        }
        MIRLSQ_LAUNCH(k_sumsq_partial<T>, dim3(nb, count), dim3(256), 0, stream, v, m, B.partials, vstride, kPartials);
        if (defer_final && !comm) { sums_pending = nb; return ok(hipGetLastError(), "sumsq"); }
        MIRLSQ_LAUNCH(k_sumsq_final<T>, dim3(count), dim3(256), 0, stream, B.partials, nb, B.sum + slot, kPartials);
        if (comm && !allreduce(B.sum + slot, (size_t)count, 2)) return false;
        return ok(hipGetLastError(), "sumsq");
    }

    // ---- row-shard exchange: sum `count` elements over the ranks, in place, ordered on the stream. kind: 0 packed
    //      [J^T J | J^T y], 1 Broyden sweep vector, 2 residual sums (mir_lsq_stats.allreduce_*)
    bool allreduce(T* buf, size_t count, int kind)
    {
        HpScope hp(this, 1);
        if (stats && !spec_enqueue) { stats->allreduce_calls[kind]++; stats->allreduce_elems[kind] += count; }
        return comm_allreduce<T>(comm, buf, count, stream) == 0;
    }

    // ---- optional per-pass trace (mir_lsq_trace)
    void trace_emit(int event, uint32_t iterations, T lambda, T residual, T trial_residual, T dx_dot)
    {
        if (!trace) return;
        if (trace->count < trace->capacity && trace->records) {
            mir_lsq_trace_record& r = trace->records[trace->count];
            r.event = event; r.iterations = iterations; r.lambda = (double)lambda; r.residual = (double)residual;
            r.trial_residual = (double)trial_residual; r.dx_dot = (double)dx_dot;
        }
        trace->count++;
    }
    // the passes of one round the reference would have executed, from the chain records and the trial sums
    bool trace_round(int ks, T residual_before, uint32_t iterations_before)
    {
        ChainRec<T> rec[kChainMax];
        T sums[kChainMax];
        if (!ok(hipMemcpyAsync(rec, B.rec, (size_t)ks * sizeof(ChainRec<T>), hipMemcpyDeviceToHost, stream), "D2H rec")
            || !ok(hipMemcpyAsync(sums, B.sum + 1, (size_t)ks * sizeof(T), hipMemcpyDeviceToHost, stream), "D2H sums")
            || !ok(hipStreamSynchronize(stream), "sync"))
            return false;
        const int dec = st_h->decision;
        if (dec == kDecideGradSmall || dec == kDecideNumericError) return true;
        for (uint32_t k = 0; k < st_h->consumed && k < (uint32_t)ks; ++k) {
            if (rec[k].flags & kFlagStepTooLong)
                trace_emit(4, iterations_before, rec[k].lambda, residual_before, 0, rec[k].new_dx_dot);
            else if ((int)k != st_h->accepted_k)
                trace_emit(2, iterations_before, rec[k].lambda, residual_before, sums[k], rec[k].new_dx_dot);
            else
                trace_emit(3, iterations_before + 1, rec[k].lambda, sums[k], sums[k], rec[k].new_dx_dot);
        }
        return true;
    }

    // wait for decision point `expect` in the pinned mirror (written by k_init_state / k_decide_chain): a spin on host
    // memory. The stream is queried now and then so that a failed kernel ends the wait instead of hanging it.
    bool wait_state(uint32_t expect)
    {
        HpScope hp(this, 3);
        st_h = st_slot[expect & 1];
        x_h = x_slot[expect & 1];
        volatile uint32_t* sq = &st_h->seq;
        // Spin briefly (a decision point is normally microseconds away), then yield the core between polls: R solver threads
        // of an in-process group, the caller's thread manager and OpenMP workers share the host. A stream that stops making
        // progress ends the wait after kWaitBoundSeconds with numericError instead of hanging the caller.
        constexpr double kWaitBoundSeconds = 600.0;
        std::chrono::steady_clock::time_point t_wait{};
        for (uint64_t spins = 0;; ++spins) {
            if (*sq == expect) break;
            if ((spins & 0xfff) == 0xfff) {
                const hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) {                       // everything enqueued has run: the image must be there
                    if (*sq == expect) break;
                    std::fprintf(stderr, "[mir_optim_amd] decision point %u was not published\n", expect);
                    return false;
                }
                if (q != hipErrorNotReady) return ok(q, "stream query");
                const auto now = std::chrono::steady_clock::now();
                if (t_wait == std::chrono::steady_clock::time_point{}) t_wait = now;
                else if (std::chrono::duration<double>(now - t_wait).count() > kWaitBoundSeconds) {
                    std::fprintf(stderr, "[mir_optim_amd] decision point %u: no progress for %.0f s, giving up\n", expect, kWaitBoundSeconds);
                    return false;
                }
            }
            if (spins < 20000) cpu_relax(); else std::this_thread::yield();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return true;
    }

    bool read_state(const T* vec_dev)
    {
        HpScope hp(this, 3);
        if (!ok(hipMemcpyAsync(st_h, B.st, sizeof(LmState<T>), hipMemcpyDeviceToHost, stream), "D2H state")) return false;
        if (vec_dev && !ok(hipMemcpyAsync(trial_h, vec_dev, n * sizeof(T), hipMemcpyDeviceToHost, stream), "D2H vec")) return false;
        return ok(hipStreamSynchronize(stream), "sync");
    }

    // ---- fused [Broyden] + J^T J + J^T y, all-reduce, unpack (LS:1003-1006, 1052, 1065)
    bool broyden_lowrank(const T* y_dev, const T* yold_dev)
    {
        T* U = static_cast<T*>(ws->ulr);
        if (lr_k >= lr_cap) {
            const uint64_t launches_before = tl_launches;
            struct Excl { Solver* s; uint64_t l0; ~Excl() { s->launches_excluded += tl_launches - l0; } } excl{this, launches_before};
            // fold the pending rank-one terms into J (the reference's successive `ger`s, LS:1006) ...
            if (!ok(lr_flush<T>(B.J, U, B.lrD, lr_k, m, (int)n, ws->num_cu, stream, Jcur == B.J ? nullptr : Jcur, Jscale), "broyden flush")) return false;
            Jcur = B.J;                          // J is materialised now
            Jscale = nullptr;
            lr_k = 0;
            if (stats) stats->broyden_flushes++;
            // ... and resynchronise: J^T J and J^T y_old recomputed from the flushed J, as the reference's syrk / gemv do
            // every pass (LS:1052, 1065). The recurrence J^T J += v dx^T + dx v^T + uu dx dx^T below then never runs for more
            // than lr_cap passes on its own rounding errors (ill-conditioned problems: cancellation could otherwise
            // accumulate over up to maxAge = 2n passes). y_old: the sweep expects J_{k-1}^T J_{k-1} in JJ; Jy is
            // rebuilt by every sweep anyway.
            if (!(variant & MIR_LSQ_VARIANT_NO_RESYNC)) {
                if (!plain_products(yold_dev)) return false;
                if (stats) stats->jtj_resyncs++;
            }
        }
        // spec_enqueue: the pass is enqueued behind the device-side guard before the host knows whether it will be needed; the
        // host-side bookkeeping (lr_k, statistics) is done when the round is committed (commit_spec_round)
        const int32_t* guard = spec_enqueue ? &B.st->spec_ok : nullptr;
        LrArgs<T> a{};
        a.J = Jcur; a.colscale = Jscale; a.U = U; a.D = B.lrD; a.dx = B.dx_acc; a.dx_dot = &B.st->dx_dot; a.y = y_dev; a.y_old = yold_dev;
        a.partials = B.lrpart; a.m = m; a.n = (int)n; a.k = lr_k; a.guard = guard;
        const int nblk = lr_blocks(m, ws->num_cu), len = lr_len((int)n);
        if (sweep_tail) {
            // the reduction -- and, with no all-reduce behind it, the n x n finish -- run in the sweep's last workgroups
            a.tail_counters = B.counters; a.range_sums = B.lrranges; a.out = B.lrvec; a.finish = comm ? 0 : 1;
            a.Dw = B.lrD; a.JJ = B.JJ; a.Jy = B.Jy; a.st = B.st;
        }
        ev_begin(1);
        if (!ok(lr_sweep<T>(a, nblk, stream), "broyden sweep")) return false;
        ev_end();
        if (!sweep_tail) MIRLSQ_LAUNCH(k_lr_reduce<T>, dim3((len + 31) / 32), dim3(32 * kReduceRanges), 0, stream, B.lrpart, nblk, len, B.lrvec, guard);
        if (comm && !allreduce(B.lrvec, (size_t)len, 1)) return false;
        if (sweep_tail && !comm) {
            // finished inside the sweep
        } else if (merge_small && (variant & MIR_LSQ_VARIANT_FINISH_IN_SOLVE)) {
            // opt-in: the solve kernel that follows applies the finish in its prologue (enqueue_solve). One launch less, but ONE
            // workgroup then does the n^2 read-modify-write that k_lr_finish spreads over n + 1: +9-11 us in the solve kernel
            // against 4.6 + 1.3 us for the kernel and its gap at n = 128 -- cfg 3 6.08 vs 6.01 ms per solve, cfg 2 equal
            finish_pending = true;
            finish_k = lr_k;
        } else {
            MIRLSQ_LAUNCH(k_lr_finish<T>, dim3(n + 1), dim3(256), 0, stream, B.lrvec, B.lrD, B.dx_acc, lr_k, (int)n, B.JJ, B.Jy, B.st, guard);
        }
        if (!spec_enqueue) {
            if (stats) stats->broyden_lr_columns += (uint64_t)lr_k;
            ++lr_k;
        }
        return ok(hipGetLastError(), "broyden finish");
    }

    // J^T J, J^T y_vec of the J in memory -> JJ, Jy (all-reduced, unpacked)
    bool plain_products(const T* y_vec)
    {
        JtjArgs<T> a{};
        a.J = B.J; a.Jout = B.J; a.y = y_vec; a.y_old = y_vec; a.dx = B.dx_acc; a.dx_dot = &B.st->dx_dot;
        a.slabs = B.slabs; a.m = m; a.n = (int)n;
        const bool direct = unpack_in_reduce(false);
        ev_begin(0);
        if (!ok(jtj_run<T>(plan, a, false, B.packed, stream, variant, direct ? unpack_target() : JtjUnpack<T>{}), "jtj kernel")) return false;
        ev_end();
        return finish_products(direct);
    }
    // MIR_LSQ_VARIANT_FD_PANEL_IS_J (opt-in): after a difference-panel refresh the panel stays the Jacobian (unscaled); the later
    // readers -- the read-only Broyden sweep and the flush -- scale on the fly. Bit-identical; measured at cfg 3: the fused
    // kernel 0.40 -> 0.37 ms without the 1 GB write, the four scaled sweeps 0.167 -> 0.182 ms each: a wash, so J is written.
    bool panel_is_J() const { return lowrank && sizeof(T) == 8 && n % 2 == 0 && (variant & MIR_LSQ_VARIANT_FD_PANEL_IS_J) != 0; }
    // Single GPU: the slab reduction writes J^T J (both triangles) and J^T y itself (the solve kernel takes |J^T y|_inf); with a
    // communicator the packed buffer is all-reduced first and k_unpack_grad expands it
    bool unpack_in_reduce(bool fd) const { return merge_small && !comm && (fd || jtj_plain_unpacks(plan)); }
    JtjUnpack<T> unpack_target()
    {
        JtjUnpack<T> u;
        u.JJ = B.JJ; u.Jy = B.Jy;
        return u;
    }
    bool finish_products(bool direct)
    {
        if (direct) return ok(hipGetLastError(), "jtj reduce");
        if (comm && !allreduce(B.packed, (size_t)n * (n + 1) / 2 + n, 0)) return false;
        MIRLSQ_LAUNCH(k_unpack_grad<T>, dim3(n + 1), dim3(128), 0, stream, B.packed, (int)n, B.JJ, B.Jy, B.st);
        return ok(hipGetLastError(), "unpack");
    }

    bool jacobian_products(bool broyden, const T* y_dev, const T* yold_dev)
    {
        if (broyden && lowrank) return broyden_lowrank(y_dev, yold_dev);
        if (!broyden) { lr_k = 0; Jcur = B.J; Jscale = nullptr; }   // J was refreshed in full: nothing is pending any more
        JtjArgs<T> a{};
        a.J = B.J; a.Jout = B.J; a.y = y_dev; a.y_old = yold_dev; a.dx = B.dx_acc; a.dx_dot = &B.st->dx_dot;
        a.slabs = B.slabs; a.m = m; a.n = (int)n;
        if (!broyden && fd_fused == 3) {
            fd_fused = 0;
            return fd_window_pipeline(y_dev);
        }
        if (!broyden && fd_fused) {
            // the row-major FD panel is still in ws->ypanel: one kernel forms the Jacobian rows (LS:1041-1047), writes
            // them to J and accumulates J^T J / J^T y from the same registers
            const bool diff = fd_fused == 2;
            fd_fused = 0;
            a.J = static_cast<const T*>(ws->ypanel); a.twh = B.twh;
            if (diff && panel_is_J()) { a.Jout = nullptr; Jcur = static_cast<const T*>(ws->ypanel); Jscale = B.twh; }
            const bool direct = unpack_in_reduce(true);
            const JtjUnpack<T> u = direct ? unpack_target() : JtjUnpack<T>{};
            ev_begin(3);
            if (!ok(diff ? jtj_run_fd_diff<T>(plan, a, B.packed, stream, u) : jtj_run_fd<T>(plan, a, B.packed, stream, u), "fd + jtj kernel")) return false;
            ev_end();
            return finish_products(direct);
        }
        const bool direct = unpack_in_reduce(false);
        ev_begin(broyden ? 1 : 0);
        if (!ok(jtj_run<T>(plan, a, broyden, B.packed, stream, variant, direct ? unpack_target() : JtjUnpack<T>{}), "jtj kernel")) return false;
        ev_end();
        return finish_products(direct);
    }

    // ---- two-stream finite-difference refresh (LS:1018-1049, 1052, 1065): the caller's batched residual kernel is MFMA-bound,
    //      the library's fused finite-difference kernel HBM-bound; in W row windows the caller's kernel for window k + 1 runs on
    //      a side stream while k_jtj_fdp consumes window k on the solver's stream (its own slab set per window: no
    //      read-modify-write of partial sums), one slab reduction over all W sets at the end.
    bool fd_window_pipeline(const T* y_dev)
    {
        if constexpr (sizeof(T) != 8) { return false; } else {
        const int W = (int)(fd_windows < (uint32_t)mir_lsq_workspace::kMaxWindows ? fd_windows : (uint32_t)mir_lsq_workspace::kMaxWindows);
        const size_t set = (size_t)plan.nblk * plan.slab_len;
        if (ws->win_slab_bytes < (size_t)W * set * sizeof(T)) {
            if (ws->win_slabs) (void)hipFree(ws->win_slabs);
            ws->win_slabs = nullptr; ws->win_slab_bytes = 0;
            if (!ok(hipMalloc(&ws->win_slabs, (size_t)W * set * sizeof(T)), "hipMalloc(window slabs)")) return false;
            ws->win_slab_bytes = (size_t)W * set * sizeof(T);
        }
        if (!ws->side_stream && !ok(hipStreamCreateWithFlags(&ws->side_stream, hipStreamNonBlocking), "side stream")) return false;
        for (int k = 0; k <= W; ++k)
            if (!ws->win_event[k] && !ok(hipEventCreateWithFlags(&ws->win_event[k], hipEventDisableTiming), "window event")) return false;
        size_t rows = (m + W - 1) / W;
        rows = (rows + 31) / 32 * 32;                    // whole 32-row stages of both kernels
        T* D = static_cast<T*>(ws->ypanel);
        if (panel_is_J()) { Jcur = D; Jscale = B.twh; }
        ev_begin(3);
        // the side stream starts once the points X (k_fd_points) are there
        if (!ok(hipEventRecord(ws->win_event[W], stream), "event") || !ok(hipStreamWaitEvent(ws->side_stream, ws->win_event[W], 0), "wait")) return false;
        int used = 0;
        for (int k = 0; k < W; ++k) {
            const size_t r0 = (size_t)k * rows;
            if (r0 >= m) break;
            const size_t rc = r0 + rows <= m ? rows : m - r0;
            fbdw(fbctx, m, n, 2 * (size_t)n, B.X, D, r0, rc, ws->side_stream);
            if (!ok(hipEventRecord(ws->win_event[k], ws->side_stream), "event") || !ok(hipStreamWaitEvent(stream, ws->win_event[k], 0), "wait")) return false;
            JtjArgs<T> a{};
            a.J = D + r0 * n; a.Jout = panel_is_J() ? nullptr : B.J + r0 * n; a.y = y_dev + r0; a.y_old = a.y; a.dx = B.dx_acc; a.dx_dot = &B.st->dx_dot;
            a.slabs = static_cast<T*>(ws->win_slabs) + (size_t)k * set; a.m = rc; a.n = (int)n; a.twh = B.twh;
            if (!ok(jtj_fdp_launch<T, false, true>(plan, a, stream), "fd window kernel")) return false;
            ++used;
        }
        if (stats) { stats->fd_callback_points += 2 * (uint64_t)n; stats->fd_callback_calls += (uint64_t)used; stats->fd_window_refreshes++; }
        JtjArgs<T> r{};
        r.slabs = static_cast<T*>(ws->win_slabs); r.n = (int)n;
        const bool direct = unpack_in_reduce(true);
        if (!ok(jtj_reduce_slabs<T>(plan, r, B.packed, stream, direct ? unpack_target() : JtjUnpack<T>{}, used * plan.nblk, plan.slab_len), "slab reduce")) return false;
        ev_end();
        return finish_products(direct);
        }
    }

    // ---- finite-difference Jacobian, device callbacks (LS:1016-1050 restructured: all perturbed
    //      points are generated at once, evaluated one by one or in batches, and written to J in
    //      coalesced column panels)
    bool fd_device()
    {
        MIRLSQ_LAUNCH(k_fd_points<T>, dim3(n), dim3(64), 0, stream, B.x, B.lower, B.upper, sd.jacobianEpsilon, (int)n, B.X, B.twh);
        for (uint32_t j = 0; j < n; ++j) {   // same arithmetic on the host, to skip collapsed intervals like LS:1033
            const T save = xh[j];
            T xmh = save - S->jacobianEpsilon, xph = save + S->jacobianEpsilon;
            xmh = std::fmax(xmh, lh[j]);
            xph = std::fmin(xph, uh[j]);
            twh_h[j] = xph - xmh;
        }
        // panel width bounded by the scratch the device can spare: half of the free HBM, at most 64 GiB (288 GB per GPU: the
        // whole 2n-point panel of cfg 4's 8e6 x 256 problem, 33 GB, stays in one piece), at least 1 GiB
        size_t pb = n;
        size_t cap = (size_t)64 << 30;
        if (ws->ypanel_bytes < 2 * pb * m * sizeof(T)) {              // the whole panel is not there yet: ask the device
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t have = ws->ypanel_bytes + free_b;         // what the panel already holds counts as available
                if (have / 2 < cap) cap = have / 2;
            }
            if (cap < ((size_t)1 << 30)) cap = (size_t)1 << 30;
        }
        if (2 * pb * m * sizeof(T) > cap) {
            pb = cap / (2 * m * sizeof(T));
            if (pb >= 64) pb -= pb % 64;         // whole multiples of 128 points per call: batched callbacks work in such chunks
        }
        if (pb < 1) pb = 1;
        if (fb && fd_batch && fd_batch / 2 < pb) pb = fd_batch / 2 ? fd_batch / 2 : 1;
        const bool no_fuse = (variant & MIR_LSQ_VARIANT_FD_SEPARATE_FILL) != 0;
        const bool use_diff = fbd && jtj_fd_diff_ok(plan, (int)n) && pb == n && sizeof(T) == 8 && !no_fuse;   // m x n difference panel
        const size_t need = (use_diff ? 1 : 2) * pb * m * sizeof(T);
        if (ws->ypanel_bytes < need) {
            if (ws->ypanel) (void)hipFree(ws->ypanel);
            ws->ypanel = nullptr; ws->ypanel_bytes = 0;
            if (!ok(hipMalloc(&ws->ypanel, need), "hipMalloc(FD panel)")) return false;
            ws->ypanel_bytes = need;
        }
        T* Y = static_cast<T*>(ws->ypanel);
        if (use_diff && fbdw && fd_windows >= 2 && plan.fdp_plain && m >= (size_t)fd_windows * 4096) {
            // two streams (jacobian_products runs the pipeline: caller's window kernels on the side stream, the fused
            // finite-difference kernel window by window on this one)
            fd_fused = 3;
            ret.fCalls += n;
            return true;
        }
        if (use_diff) {
            // all 2n points in one sweep, the caller's kernel hands over D[i][j] = f(x + h e_j)_i - f(x - h e_j)_i (LS:1041, 1045);
            // k_jtj_fdp<., false, true> (jacobian_products) scales the columns (LS:1047), writes J and accumulates J^T J / J^T y
            ev_begin(4);
            fbd(fbctx, m, n, 2 * (size_t)n, B.X, Y);
            ev_end();
            if (stats) { stats->fd_callback_points += 2 * (uint64_t)n; stats->fd_callback_calls++; }
            fd_fused = 2;
            ret.fCalls += n;
            return ok(hipGetLastError(), "fd batched callback");
        }
        if (fbr && (plan.fdp || plan.fdp8) && pb == n && sizeof(T) == 8 && !no_fuse) {
            // all 2n points in one sweep, Y[i][2j], Y[i][2j+1] = f(x + h e_j)_i, f(x - h e_j)_i; k_jtj2<., false, true>
            // (jacobian_products) turns the pairs into Jacobian rows on its way to J^T J -- no k_fd_fill pass
            ev_begin(4);
            fbr(fbctx, m, n, 2 * (size_t)n, B.X, Y);
            ev_end();
            if (stats) { stats->fd_callback_points += 2 * (uint64_t)n; stats->fd_callback_calls++; }
            fd_fused = 1;
            ret.fCalls += n;
            return ok(hipGetLastError(), "fd batched callback");
        }
        for (size_t j0 = 0; j0 < n; j0 += pb) {
            const size_t pc = (j0 + pb <= n) ? pb : n - j0;
            ev_begin(4);
            if (fb) {
                fb(fbctx, m, n, 2 * pc, B.X + 2 * j0 * n, Y);
                if (stats) stats->fd_callback_points += 2 * (uint64_t)pc;
            } else {
                for (size_t c = 0; c < pc; ++c) {
                    if (twh_h[j0 + c] == 0) continue;
                    f(fctx, m, n, B.X + (2 * (j0 + c)) * n, Y + (2 * c) * m);
                    f(fctx, m, n, B.X + (2 * (j0 + c) + 1) * n, Y + (2 * c + 1) * m);
                    if (stats) stats->fd_callback_points += 2;
                }
            }
            ev_end();
            if (stats) stats->fd_callback_calls++;       // one timed bracket per panel (as the events count them)
            dim3 grid((unsigned)((m + 63) / 64), (unsigned)((pc + 31) / 32));
            MIRLSQ_LAUNCH(k_fd_fill<T>, grid, dim3(256), 0, stream, Y, m, B.twh, B.J, m, (int)n, (int)j0, (int)pc);
        }
        ret.fCalls += n;    // LS:1024, LS:1049 (quirk Q5: +n although 2n evaluations are made)
        return ok(hipGetLastError(), "fd fill");
    }

    // ---- finite-difference Jacobian, host callbacks: the reference's task body LS:1019-1048
    static void fd_task_trampoline(mir_least_squares_task task, uint32_t totalThreads, uint32_t threadId, uint32_t j)
    {
        static_cast<Solver<T>*>(task.context)->fd_task(totalThreads, threadId, j);
    }
    void fd_task(uint32_t totalThreads, uint32_t threadId, uint32_t j)
    {
        const uint32_t idx = totalThreads >= n ? j : threadId;       // LS:1022
        if (idx >= n || j >= n) { fd_failed = true; return; }
        ++fd_tasks_run;
        // the manager's worker threads start on device 0: select the solver's device before any runtime call
        if (hipSetDevice(device) != hipSuccess) { fd_failed = true; return; }
        Slot* s;
        {
            std::lock_guard<std::mutex> lk(fd_mutex);
            s = &slots[idx];
            if (!s->p) {
                s->p = static_cast<T*>(std::malloc(n * sizeof(T)));
                if (!s->p) { fd_failed = true; return; }
            }
            if (!fd_panel_mode && !s->yp) {
                if (hipHostMalloc((void**)&s->yp, m * sizeof(T), hipHostMallocDefault) != hipSuccess
                    || hipHostMalloc((void**)&s->ym, m * sizeof(T), hipHostMallocDefault) != hipSuccess) {
                    fd_failed = true;
                    return;
                }
            }
            if (slot_count[idx]++ == 0) std::memcpy(s->p, xh, n * sizeof(T));   // LS:1024-1025
        }
        T* p = s->p;
        const T save = p[j];                                         // LS:1027-1031
        T xmh = save - S->jacobianEpsilon, xph = save + S->jacobianEpsilon;
        xmh = std::fmax(xmh, lh[j]);
        xph = std::fmin(xph, uh[j]);
        const T twh = xph - xmh;
        if (fd_panel_mode) {
            // The caller's f writes f(x + h e_j), f(x - h e_j) straight into rows 2j, 2j + 1 of the pinned point-major panel;
            // ONE asynchronous copy takes the pair to the device panel on a copy stream while this thread -- and the
            // manager's other threads -- go on with the next column. No lock, no synchronisation: fd_host() makes the
            // solver's stream wait for the copy streams once, at the end. (A collapsed interval, LS:1033, evaluates and
            // copies nothing: k_fd_fill writes the zero column without reading the panel.)
            if (twh != 0) {
                T* hp = static_cast<T*>(ws->pinned_panel) + (size_t)(2 * j) * m;
                const auto t0 = std::chrono::steady_clock::now();
                p[j] = xph;
                f(fctx, m, n, p, hp);                                // LS:1035-1036
                p[j] = xmh;
                f(fctx, m, n, p, hp + m);                            // LS:1038-1039
                p[j] = save;                                         // LS:1040
                fd_f_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                const int k = (int)(idx % mir_lsq_workspace::kCopyStreams);
                fd_streams_used |= 1u << k;
                if (hipMemcpyAsync(static_cast<T*>(ws->ypanel) + (size_t)(2 * j) * m, hp, 2 * m * sizeof(T), hipMemcpyHostToDevice,
                                   ws->copy_stream[k]) != hipSuccess)
                    fd_failed = true;
            }
            return;
        }
        if (twh != 0) {                                              // LS:1033-1043
            p[j] = xph;
            f(fctx, m, n, p, s->yp);
            p[j] = xmh;
            f(fctx, m, n, p, s->ym);
            p[j] = save;
        }
        // Staging: the CURRENT mBuffer (it holds y_old, which is dead while the Jacobian is refreshed in full: the next
        // Broyden update only comes after another accepted step has rewritten it) and the free m-vector `fr` -- never the
        // live residual `y`, whichever of the three physical buffers it is in after the rotations of the accepted steps.
        std::lock_guard<std::mutex> lk(fd_mutex);
        if (twh != 0) {
            if (hipMemcpyAsync(mB, s->yp, m * sizeof(T), hipMemcpyHostToDevice, stream) != hipSuccess
                || hipMemcpyAsync(fr, s->ym, m * sizeof(T), hipMemcpyHostToDevice, stream) != hipSuccess) {
                fd_failed = true;
                return;
            }
        }
        MIRLSQ_LAUNCH(k_fd_fill_col<T>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream,
                           mB, fr, twh, B.J, m, (int)n, (int)j);
        if (hipStreamSynchronize(stream) != hipSuccess) fd_failed = true;
    }
    // Can this refresh stage through the pinned point-major panel? Needs the whole 2n x m panel on both sides: device (the
    // budget fd_device() uses: half of the free HBM, at most 64 GiB) and pinned host memory (at most 16 GiB). Otherwise the
    // column-at-a-time path below (per-slot staging vectors, one strided column write per task) serves any size.
    bool fd_host_prepare_panel()
    {
        if (variant & MIR_LSQ_VARIANT_FD_HOST_COLUMNS) return false;
        const size_t need = 2 * (size_t)n * m * sizeof(T);
        if (need > ((size_t)16 << 30)) return false;
        if (ws->ypanel_bytes < need) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
            size_t cap = (ws->ypanel_bytes + free_b) / 2;
            if (cap > ((size_t)64 << 30)) cap = (size_t)64 << 30;
            if (need > cap) return false;
            if (ws->ypanel) (void)hipFree(ws->ypanel);
            ws->ypanel = nullptr; ws->ypanel_bytes = 0;
            if (hipMalloc(&ws->ypanel, need) != hipSuccess) return false;
            ws->ypanel_bytes = need;
        }
        if (ws->pinned_panel_bytes < need) {
            if (ws->pinned_panel) (void)hipHostFree(ws->pinned_panel);
            ws->pinned_panel = nullptr; ws->pinned_panel_bytes = 0;
            if (hipHostMalloc(&ws->pinned_panel, need, hipHostMallocDefault) != hipSuccess) return false;
            ws->pinned_panel_bytes = need;
        }
        for (int k = 0; k < mir_lsq_workspace::kCopyStreams; ++k) {
            if (!ws->copy_stream[k] && hipStreamCreateWithFlags(&ws->copy_stream[k], hipStreamNonBlocking) != hipSuccess) return false;
            if (!ws->copy_event[k] && hipEventCreateWithFlags(&ws->copy_event[k], hipEventDisableTiming) != hipSuccess) return false;
        }
        return true;
    }
    bool fd_host()
    {
        slots.resize(n);
        slot_count.assign(n, 0);                                     // LS:1018
        fd_panel_mode = fd_host_prepare_panel();
        fd_streams_used = 0;
        fd_tasks_run = 0;
        if (fd_panel_mode) {
            // twh (and the points, unused here) on the device with the arithmetic of LS:1027-1031: k_fd_fill needs the widths
            MIRLSQ_LAUNCH(k_fd_points<T>, dim3(n), dim3(64), 0, stream, B.x, B.lower, B.upper, sd.jacobianEpsilon, (int)n, B.X, B.twh);
        }
        mir_least_squares_task task{this, nullptr};
        const auto t0 = std::chrono::steady_clock::now();
        if (tm) tm(tmctx, n, task, &fd_task_trampoline);             // LS:1019
        else for (uint32_t j = 0; j < n; ++j) fd_task(1, 0, j);      // LS:947-951
        uint32_t calls = 0;
        for (uint32_t k = 0; k < n; ++k) calls += (uint32_t)slot_count[k];
        ret.fCalls += calls;                                         // LS:1049
        if (stats) {
            stats->fd_host_wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            stats->fd_host_f_ms += (double)fd_f_ns.exchange(0) * 1e-6;
            stats->fd_host_columns += n;
        }
        const bool incomplete = fd_tasks_run.load() != n;
        if (incomplete)     // a manager that stops half way (an exception in a binding, a cancelled pool) leaves columns of J stale: fail loudly
            std::fprintf(stderr, "[mir_optim_amd] thread manager ran %u of %u finite-difference tasks\n", fd_tasks_run.load(), n);
        if (incomplete || fd_failed) {
            // copies of the columns that did get evaluated may still be reading the pinned panel: let them finish before the
            // caller (or an owned workspace's teardown) can touch it
            if (fd_panel_mode)
                for (int k = 0; k < mir_lsq_workspace::kCopyStreams; ++k)
                    if (fd_streams_used.load() & (1u << k)) (void)hipStreamSynchronize(ws->copy_stream[k]);
            return false;
        }
        if (fd_panel_mode) {
            // the solver's stream waits for the copy streams (no host synchronisation), then ONE coalesced conversion of the
            // whole panel: pairs of m-vectors -> column panels of J through an LDS transpose (LS:1041-1047)
            const uint32_t used = fd_streams_used.load();
            for (int k = 0; k < mir_lsq_workspace::kCopyStreams; ++k) {
                if (!(used & (1u << k))) continue;
                if (!ok(hipEventRecord(ws->copy_event[k], ws->copy_stream[k]), "copy event")
                    || !ok(hipStreamWaitEvent(stream, ws->copy_event[k], 0), "wait for the panel copies")) return false;
            }
            dim3 grid((unsigned)((m + 63) / 64), (unsigned)((n + 31) / 32));
            MIRLSQ_LAUNCH(k_fd_fill<T>, grid, dim3(256), 0, stream, static_cast<const T*>(ws->ypanel), m, B.twh, B.J, m, (int)n, 0, (int)n);
            return ok(hipGetLastError(), "fd fill");
        }
        return true;
    }

    bool analytic_jacobian()
    {
        if (device_cb) {
            g(gctx, m, n, B.x, B.J);
        } else {
            if (!ws->pinned_J && !ok(hipHostMalloc(&ws->pinned_J, m * n * sizeof(T), hipHostMallocDefault), "hipHostMalloc(J)")) return false;
            T* Jh = static_cast<T*>(ws->pinned_J);
            g(gctx, m, n, xh, Jh);
            if (!ok(hipMemcpyAsync(B.J, Jh, m * n * sizeof(T), hipMemcpyHostToDevice, stream), "H2D J")
                || !ok(hipStreamSynchronize(stream), "sync")) return false;
        }
        ret.gCalls += 1;                                             // LS:1014
        return true;
    }

    // the n x n part of a round (LS:1053-1110, 1141-1142) for ks ladder entries
    bool enqueue_solve(int ks, const T* lam, bool check_grad, bool lambda_from_state)
    {
        LmSolveArgs<T> a{};
        a.JJ = B.JJ; a.Jy = B.Jy; a.x = B.x; a.lower = B.lower; a.upper = B.upper;
        a.dx = B.dx; a.trial = B.trial; a.st = B.st; a.rec = B.rec; a.set = sd; a.n = (int)n;
        for (int k = 0; k < kChainMax; ++k) { a.sc[k] = B.sc[k]; a.lam[k] = (lam && k < ks) ? lam[k] : T(0); }
        a.f_in_lds = f_in_lds;
        a.check_grad = check_grad ? 1 : 0;
        a.lambda_from_state = lambda_from_state ? 1 : 0;
        a.lambda_from_device = spec_enqueue ? 1 : 0;
        a.guard = spec_enqueue ? &B.st->spec_ok : nullptr;
        if (finish_pending) {
            if (ks != 1) { std::fprintf(stderr, "[mir_optim_amd] internal error: a Broyden finish is pending for a ladder of %d\n", ks); return false; }
            a.lr = B.lrvec; a.lrD = B.lrD; a.lr_dx = B.dx_acc; a.lr_k = finish_k; a.JJw = B.JJ; a.Jyw = B.Jy;
            finish_pending = false;
        }
        if (!dbg_solve) a.sc[0].dbg = nullptr;
        ev_begin(2);
        {
            HpScope hp(this, 4);
            if (!ok(launch_solve(a, ks), "solve launch")) return false;
        }
        ev_end();
        return true;
    }

    // the decision of a round (LS:1080-1161) for ks trials whose sums of squares are in B.sum + 1; publishes decision point ++seq
    //      sum_v != nullptr: the ks trial residual vectors at sum_v + k m still have to be summed -- on a single GPU the sweep
    //      and the decision are one launch (k_sumsq_tail<., kSumsqTailDecide>), otherwise sumsq() runs first
    bool enqueue_decide(int ks, bool check_grad, bool lambda_from_state, bool next_round_enqueued_ahead, const T* sum_v = nullptr)
    {
        const bool one_launch = sum_v && sumsq_tail && !comm;
        if (sum_v && !one_launch && !sumsq(sum_v, 1, ks, m, true)) return false;
        DecideArgs<T> d{};
        d.sums = B.sum + 1; d.rec = B.rec; d.st = B.st; d.set = sd; d.x = B.x; d.trial = B.trial; d.dx_chain = B.dx;
        d.dx_acc = B.dx_acc; d.n = (int)n; d.ks = ks; d.check_grad = check_grad ? 1 : 0;
        d.lambda_from_state = lambda_from_state ? 1 : 0;
        ++seq;
        d.host_st = st_slot_d[seq & 1]; d.host_x = x_slot_d[seq & 1]; d.seq = seq;
        d.guard = spec_enqueue ? &B.st->spec_ok : nullptr;
        d.spec_static = next_round_enqueued_ahead ? 1 : 0;
        d.maxIterations = S->maxIterations;
        d.partials = B.partials; d.nparts = sums_pending; d.pstride = kPartials;
        sums_pending = 0;
        if (one_launch) {
            SumsqTailArgs<T> t = sumsq_tail_args(sum_v, 1, m);
            t.dec = d;
            MIRLSQ_LAUNCH((k_sumsq_tail<T, kSumsqTailDecide>), dim3(sumsq_blocks(), ks), dim3(kSolveThreads), 0, stream, t);
        } else {
            MIRLSQ_LAUNCH(k_decide_chain<T>, dim3(1), dim3(kSolveThreads), 0, stream, d);
        }
        return ok(hipGetLastError(), "decide kernel");
    }

    // Enqueue, behind the guard, the LIBRARY part of the round that follows an ACCEPTED round with one trial: the Broyden
    // sweep with the roles the residual buffers will have after the rotation (y_new = fr, y_old = y) and the solve with
    // lambda taken from the device state. The caller's residual callback is NOT enqueued ahead of time: it cannot be guarded
    // (a full sweep over the caller's data per miss -- measured at cfg 3: two misses per solve cost more than the launch
    // latency the scheme hides); the host enqueues it, the sum of squares and the decision once the round is committed,
    // while the GPU is busy with the sweep and the solve. The host-side bookkeeping waits for commit_spec_round().
    bool enqueue_spec_round()
    {
        spec_enqueue = true;
        spec_events_from = events.size();
        const bool good = broyden_lowrank(fr, y) && enqueue_solve(1, nullptr, true, false);
        spec_enqueue = false;
        return good;
    }
    // the round enqueued ahead of time is the one the reference runs next: do now what the host does when it enqueues a
    // Broyden round itself
    void commit_spec_round()
    {
        for (size_t i = spec_events_from; i < events.size(); ++i) if (events[i].kind >= 100) events[i].kind -= 100;
        if (stats) {
            stats->jacobian_broyden++;
            stats->broyden_lr_columns += (uint64_t)lr_k;
            if (comm) { stats->allreduce_calls[1]++; stats->allreduce_elems[1] += (uint64_t)lr_len((int)n); }
        }
        ++lr_k;
    }
    void drop_spec_round()
    {
        for (size_t i = spec_events_from; i < events.size(); ++i) if (events[i].kind >= 100) events[i].kind = -1;
    }

    Result run()
    {
        ret.status = mir_ls_numericError; ret.iterations = 0; ret.fCalls = 0; ret.gCalls = 0;   // LS:132-142
        ret.residual = Lim<T>::inf(); ret.lambda = 0;
        const auto t_start = std::chrono::steady_clock::now();
        launches_mark = tl_launches;

        // validation, LS:930-943 (quirk Q9) -- needs no device
        {
            bool finite = true;
            for (uint32_t i = 0; i < n; ++i) if (!(-Lim<T>::inf() < xh[i] && xh[i] < Lim<T>::inf())) finite = false;
            if (m == 0 || n == 0 || !finite) { ret.status = mir_ls_badGuess; return ret; }
            for (uint32_t i = 0; i < n; ++i) if (!(lh[i] <= xh[i]) || !(xh[i] <= uh[i])) { ret.status = mir_ls_badBounds; return ret; }
            if (!(0 <= S->minStepQuality && S->minStepQuality < 1)) { ret.status = mir_ls_badMinStepQuality; return ret; }
            if (!(0 <= S->goodStepQuality && S->goodStepQuality <= 1)) { ret.status = mir_ls_badGoodStepQuality; return ret; }
            if (!(S->minStepQuality < S->goodStepQuality)) { ret.status = mir_ls_badStepQuality; return ret; }
            if (!(1 <= S->lambdaIncrease && S->lambdaIncrease <= std::sqrt(Lim<T>::max))) { ret.status = mir_ls_badLambdaParams; return ret; }
            if (!(std::sqrt(Lim<T>::min_normal) <= S->lambdaDecrease && S->lambdaDecrease <= 1)) { ret.status = mir_ls_badLambdaParams; return ret; }
        }
        if (!device_available()) return ret;
        if (!setup()) { teardown(); return ret; }
        has_bounds = (variant & MIR_LSQ_VARIANT_SOLVE_BOUNDED) != 0;
        for (uint32_t i = 0; i < n; ++i)
            if (lh[i] > -Lim<T>::inf() || uh[i] < Lim<T>::inf()) has_bounds = true;

        const uint32_t maxAge = S->maxAge ? S->maxAge : (g ? 3 : 2 * n);     // LS:945 (quirk Q4)

        bool fail = false;
        do {   // single-exit block for device errors
            if (!eval_f(B.x, xh, y)) { fail = true; break; }                 // LS:953
            ++ret.fCalls;
            ++seq;
            if (sumsq_tail && !comm) {                                       // LS:955 + the state at entry, one launch
                SumsqTailArgs<T> t = sumsq_tail_args(y, 0, 0);
                t.st = B.st; t.host_st = st_slot_d[seq & 1]; t.seq = seq;
                MIRLSQ_LAUNCH((k_sumsq_tail<T, kSumsqTailInit>), dim3(sumsq_blocks(), 1), dim3(kSolveThreads), 0, stream, t);
            } else {
                if (!sumsq(y, 0)) { fail = true; break; }                    // LS:955
                MIRLSQ_LAUNCH(k_init_state<T>, dim3(1), dim3(1), 0, stream, B.sum, B.st, st_slot_d[seq & 1], seq);
            }
            if (!ok(hipGetLastError(), "init state") || !wait_state(seq)) { fail = true; break; }
        } while (false);
        if (fail) { teardown(); ret.status = mir_ls_numericError; return ret; }

        ret.residual = st_h->residual;
        bool fConverged = ret.residual <= S->maxGoodResidual;                // LS:956
        bool needJacobian = true;                                            // LS:959
        bool last_rejected = false;
        bool spec_live = false;            // the round at the top of the loop is already enqueued (guard open)
        bool spec_predict = true;          // the last first trial after a Jacobian update was accepted
        const bool speculate = device_cb && !no_speculation;        // ladder trials: one fb call, or ks calls of f
        uint32_t age = maxAge;
        ret.lambda = 0;
        T mu = 1;
        const T suspiciousMu = 16;
        ret.status = mir_ls_maxIterations;                                   // LS:971

        do {
            close_round();
            if (stats) stats->passes++;
            if (fConverged) { ret.status = mir_ls_fConverged; break; }       // LS:974-978
            if (!(ret.lambda <= S->maxLambda)) { ret.status = mir_ls_furtherImprovement; break; }   // LS:979-983
            if (mu > suspiciousMu && age) {                                  // LS:984-989
                needJacobian = true;
                age = maxAge;
                mu = 1;
                MIRLSQ_LAUNCH(k_reset_mu<T>, dim3(1), dim3(1), 0, stream, B.st);
            }
            {                                                                // LS:990-995
                bool nan = false;
                for (uint32_t i = 0; i < n; ++i) if (!(xh[i] <= xh[i])) nan = true;
                if (nan) { ret.status = mir_ls_numericError; break; }
            }
            bool newJacobian = false;
            int ks = 1;
            bool lambda_from_state = false, skip_eval = false;
            T* ytr = fr;
            bool solve_enqueued = false;
            if (spec_live) {
                // this round is already in the stream (enqueue_spec_round of the previous iteration) and its guard is open
                spec_live = false;
                if (!needJacobian || !(age < maxAge) || lr_k >= lr_cap) {
                    std::fprintf(stderr, "[mir_optim_amd] internal error: the round enqueued ahead of time is not the next round\n");
                    fail = true;
                    break;
                }
                needJacobian = false;
                newJacobian = true;
                last_rejected = false;
                age++;
                commit_spec_round();
                solve_enqueued = true;
            }
            round_kind = solve_enqueued ? 1 : 2;
            if (needJacobian) {                                              // LS:996-1063
                needJacobian = false;
                newJacobian = true;
                last_rejected = false;
                round_kind = age < maxAge ? 1 : 0;
                if (age < maxAge) {                                          // Broyden, LS:999-1007
                    age++;
                    if (stats) stats->jacobian_broyden++;
                    if (!jacobian_products(true, y, mB)) { fail = true; break; }
                    trace_emit(1, ret.iterations, ret.lambda, ret.residual, 0, st_h->dx_dot);
                } else {
                    age = 0;
                    if (stats) stats->jacobian_full++;
                    const auto t0 = std::chrono::steady_clock::now();
                    bool okj;
                    if (g) okj = analytic_jacobian();                        // LS:1011-1015
                    else okj = device_cb ? fd_device() : fd_host();          // LS:1016-1050
                    if (!okj) { fail = true; break; }
                    // fd_ms: host-callback mode is synchronous anyway (wall clock); in device-callback mode the refresh is only
                    // ENQUEUED here -- no stream synchronisation for the sake of a statistic: fd_callback_ms (events) covers it
                    if (stats && !device_cb)
                        stats->fd_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    if (!jacobian_products(false, y, mB)) { fail = true; break; }
                    trace_emit(0, ret.iterations, ret.lambda, ret.residual, 0, st_h->dx_dot);
                }
            }

            // ---- one ROUND: the n x n solve for a ladder of lambdas, the trial residuals, the decision.
            // LS:1053-1062 (gradient test, inside the kernel when a new Jy exists), LS:1067-1110, 1141-1142
            // (damping, BOXCQP, step rounding, trial point, prediction), LS:1112-1161 (trial residual, acceptance).
            //
            // Speculation: after a rejection the reference re-solves with lambda * lambdaIncrease * mu, mu * 2
            // (LS:1103, 1127) and J^T J, J^T y unchanged -- the whole ladder lambda_0 .. lambda_{ks-1} is known in
            // advance. Workgroup k solves with lambda_k, all trial points are evaluated (one sweep of the batched
            // residual callback, or one call of f per point) and k_decide_chain walks them in the reference's order; entries after the
            // first accepted one are discarded, so results, counters and callback-visible semantics of accepted
            // points are unchanged. The ladder stops where the reference's top-of-loop checks would intervene
            // (lambda > maxLambda LS:979, forced refresh LS:984).
            lambda_from_state = !solve_enqueued && !(ret.lambda >= S->minLambda);   // first pass: lambda_0 rule inside the kernel
            T lam[kChainMax];
            lam[0] = ret.lambda;
            if (speculate && !newJacobian && !lambda_from_state && last_rejected) {
                T l2 = ret.lambda, m2 = mu;
                while (ks < kChainMax) {
                    l2 *= S->lambdaIncrease * m2;
                    m2 *= 2;
                    if (!(l2 <= S->maxLambda)) break;                         // LS:979 would exit there
                    if (m2 > suspiciousMu && age) break;                      // LS:984 would force a refresh there
                    lam[ks++] = l2;
                }
            }
            if (!solve_enqueued && !enqueue_solve(ks, lam, newJacobian, lambda_from_state)) { fail = true; break; }
            if (dbg_solve) {
                long long h[32];
                if (hipMemcpy(h, B.sc[0].dbg, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
                    std::fprintf(stderr, "[solve dbg] (10ns ticks) build %lld  copy/equil %lld  scale %lld  potrf %lld  potrs %lld  refine %lld  epilogue %lld  total %lld  shader MHz %.0f  [matvec1 %lld berr %lld]\n",
                                 h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[8] - h[7], h[8] - h[0], (double)(h[10] - h[9]) / (double)(h[8] - h[0]) * 100.0, h[11] - h[5], h[12] - h[11]);
                    if (n > 128 && n <= (uint32_t)kSolveMaxN)
                        std::fprintf(stderr, "[solve dbg] potrf_panel steps (10ns ticks, summed over the panels): earlier panels on MFMA %lld  diagonal rows %lld  other rows + store %lld\n", h[16], h[17], h[18]);
                }
            }

            // null-step probe: one small read-back instead of ks residual evaluations, only while the tail is running
            if (device_cb && tail_null && last_rejected && !newJacobian && !no_null_skip) {
                HpScope hp(this, 3);
                ChainRec<T> rr[kChainMax];
                if (!ok(hipMemcpyAsync(rr, B.rec, (size_t)ks * sizeof(ChainRec<T>), hipMemcpyDeviceToHost, stream), "D2H rec")
                    || !ok(hipStreamSynchronize(stream), "sync")) { fail = true; break; }
                skip_eval = true;
                for (int k = 0; k < ks; ++k) if (!(rr[k].flags & kFlagNullStep)) skip_eval = false;
                if (skip_eval && stats) stats->elided_evaluations += (uint64_t)ks;
            }

            // trial residuals -> ytr (k-th vector at ytr + k * m); with one trial they go straight into the free buffer
            if (ks > 1) ytr = static_cast<T*>(ws->ytrial);
            if (skip_eval) {
                // every trial of the round equals x: the decision kernel substitutes the residual it already has
            } else if (device_cb) {
                // no host round trip before the residual: it is evaluated speculatively even when the record will
                // forbid it (gradient converged, QP failure, step guard) -- the decision kernel then ignores it
                HpScope hp(this, 2);
                ev_begin(5);
                if (ks > 1 && fb) fb(fbctx, m, n, (size_t)ks, B.trial, ytr);
                else for (int k = 0; k < ks; ++k) f(fctx, m, n, B.trial + (size_t)k * n, ytr + (size_t)k * m);
                ev_end();
                if (stats) { stats->trial_callback_points += (uint64_t)ks; stats->trial_callback_calls++; }
            } else {
                // reference contract: the callback needs the trial point on the host
                ChainRec<T> r0;
                if (!ok(hipMemcpyAsync(&r0, B.rec, sizeof r0, hipMemcpyDeviceToHost, stream), "D2H rec") || !read_state(B.trial)) { fail = true; break; }
                const bool null_step = (r0.flags & kFlagNullStep) && !no_null_skip;     // trial_h == xh bit for bit
                if (null_step && stats) stats->elided_evaluations++;
                const bool no_f = (newJacobian && (r0.flags & kFlagGradSmall)) || r0.qp_status != 0
                    || (r0.flags & (kFlagDxNaN | kFlagStepTooLong)) || null_step;
                if (!no_f && !eval_f(B.trial, trial_h, ytr)) { fail = true; break; }
            }

            // Can the round after this one be enqueued before this one's decision is known? Only the common case is covered:
            // one trial now, and -- if it is accepted and no exit test fires (decided on the device, k_decide_chain) -- a
            // Broyden pass next that needs neither a full refresh (age) nor a flush of the pending terms (lr_k).
            const bool decide_static = pipeline && spec_predict && ks == 1 && !skip_eval && age < maxAge && lr_k < lr_cap;
            if (!enqueue_decide(ks, newJacobian, lambda_from_state, decide_static, skip_eval ? nullptr : ytr)) { fail = true; break; }
            const uint32_t round_seq = seq;
            if (decide_static && !enqueue_spec_round()) { fail = true; break; }
            if (!wait_state(round_seq)) { fail = true; break; }
            if (decide_static) {
                if (st_h->spec_ok) spec_live = true; else drop_spec_round();
            }
            // one-bit predictor: enqueue ahead only while first trials are being accepted (the rejection tail of a noisy fit
            // would waste a guarded round per miss)
            if (ks == 1 && newJacobian) spec_predict = st_h->decision == kDecideAccept;

            if (trace && !trace_round(ks, ret.residual, ret.iterations)) { fail = true; break; }
            const int dec = st_h->decision;
            tail_null = st_h->null_tail != 0;
            ret.fCalls += st_h->fcalls;                                      // LS:1112
            if (stats) {
                if (st_h->consumed > 1) stats->passes += st_h->consumed - 1;
                stats->rejected += st_h->rejects;
                stats->step_guard_rejects += st_h->guards;
                stats->qp_active_set_passes += st_h->qp_active;
            }
            if (dec == kDecideGradSmall) {                                   // LS:1053-1062
                if (age == 0) { ret.status = mir_ls_gConverged; break; }
                age = maxAge;
                continue;
            }
            ret.lambda = st_h->lambda;
            mu = st_h->mu;
            if (dec == kDecideNumericError) { ret.status = mir_ls_numericError; break; }   // LS:1080-1092, 1117-1122
            if (dec == kDecideReject) { last_rejected = true; continue; }    // LS:1101-1106, 1125-1130
            last_rejected = false;

            needJacobian = true;                                             // LS:1132-1139
            ret.iterations = st_h->iterations;
            for (uint32_t i = 0; i < n; ++i) xh[i] = x_h[i];                 // the decision kernel published the new x
            if (ytr != fr) {
                if (!ok(hipMemcpyAsync(fr, ytr + (size_t)st_h->accepted_k * m, m * sizeof(T), hipMemcpyDeviceToDevice, stream), "D2D y")) { fail = true; break; }
            }
            { T* t = mB; mB = y; y = fr; fr = t; }                           // swap(mBuffer, y) of LS:1136 as a rotation of three
            ret.residual = st_h->residual;
            fConverged = ret.residual <= S->maxGoodResidual;
            if (stats) stats->accepted++;

            if (dec == kDecideAcceptNoPrediction) { ret.status = mir_ls_furtherImprovement; break; }   // LS:1144-1148

            const T dxn = std::sqrt(st_h->dx_dot);                           // LS:1164-1173 (quirk Q6)
            if (!(dxn > S->absTolerance && st_h->trial_xnorm > dxn * S->relTolerance)) {
                if (age == 0) { ret.status = mir_ls_xConverged; break; }
                age = maxAge;
                continue;
            }
        } while (ret.iterations < S->maxIterations);                         // LS:1175

        close_round();
        if (fail) ret.status = mir_ls_numericError;
        if (stats) stats->total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        teardown();
        return ret;
    }
};

template <typename T>
typename Abi<T>::Result solve_entry(const typename Abi<T>::Settings* settings, size_t m, size_t n, T* x, const T* l,
                                    const T* u, const mir_lsq_gpu_options* opt, void* fctx, typename Abi<T>::F f,
                                    void* gctx, typename Abi<T>::G g, void* tmctx, mir_least_squares_thread_manager tm)
{
    Solver<T> s{};
    s.S = settings; s.m = m; s.n = (uint32_t)n; s.xh = x; s.lh = l; s.uh = u;
    s.fctx = fctx; s.f = f; s.gctx = gctx; s.g = g; s.tmctx = tmctx; s.tm = tm;
    if (opt) {
        s.device_cb = (opt->flags & MIR_LSQ_DEVICE_CALLBACKS) != 0;
        s.time_kernels = (opt->flags & MIR_LSQ_TIME_KERNELS) != 0 && opt->stats;
        s.stream = static_cast<hipStream_t>(opt->stream);
        s.comm = opt->comm;
        s.ws = opt->workspace;
        s.fbctx = opt->fbContext;
        s.fb = s.device_cb ? reinterpret_cast<typename Abi<T>::FB>(opt->fb) : nullptr;
        s.fd_batch = opt->fd_batch;
        s.variant = opt->variant;
        if (opt->stats) {
            // mir_lsq_stats is versioned by size (header: "Versioning of mir_lsq_stats"): work on a full local image, hand back
            // only what the caller's struct holds
            size_t bytes = opt->struct_size >= offsetof(mir_lsq_gpu_options, fbRowMajorDiff) + sizeof(void*)
                ? offsetof(mir_lsq_stats, trial_callback_points) + sizeof(uint64_t)
                : (opt->struct_size >= offsetof(mir_lsq_gpu_options, fbRowMajor) + sizeof(void*)
                       ? offsetof(mir_lsq_stats, jtj_fd_launches) + sizeof(uint64_t)
                       : offsetof(mir_lsq_stats, qp_active_set_passes) + sizeof(uint64_t));
            if (opt->struct_size >= offsetof(mir_lsq_gpu_options, stats_size) + sizeof(uint32_t) && opt->stats_size)
                bytes = opt->stats_size;
            if (bytes > sizeof(mir_lsq_stats)) bytes = sizeof(mir_lsq_stats);
            s.stats_user = opt->stats;
            s.stats_bytes = bytes;
            std::memcpy(&s.stats_local, opt->stats, bytes);          // the counters accumulate over calls
            s.stats = &s.stats_local;
        }
        if (opt->struct_size >= offsetof(mir_lsq_gpu_options, trace) + sizeof(void*)) s.trace = opt->trace;
        if (opt->struct_size >= offsetof(mir_lsq_gpu_options, fbRowMajor) + sizeof(void*) && s.device_cb)
            s.fbr = reinterpret_cast<typename Abi<T>::FB>(opt->fbRowMajor);
        if (opt->struct_size >= offsetof(mir_lsq_gpu_options, fbRowMajorDiff) + sizeof(void*) && s.device_cb)
            s.fbd = reinterpret_cast<typename Abi<T>::FB>(opt->fbRowMajorDiff);
        if (opt->struct_size >= offsetof(mir_lsq_gpu_options, fbRowMajorDiffWindow) + sizeof(void*) && s.device_cb && sizeof(T) == 8) {
            s.fbdw = reinterpret_cast<mir_lsq_window_function_d>(opt->fbRowMajorDiffWindow);
            s.fd_windows = opt->fd_windows;
        }
        if (s.trace) s.trace->count = 0;
    }
    const typename Abi<T>::Result r = s.run();
    if (s.stats_user) std::memcpy(s.stats_user, &s.stats_local, s.stats_bytes);
    return r;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

size_t mir_box_qp_work_length(size_t n) { return n * n * 2 + n * 8; }                                  // QP:36-42
size_t mir_box_qp_iwork_length(size_t n) { return n + (n / sizeof(int32_t) + (n % sizeof(int32_t) != 0)); }  // QP:47-50
size_t mir_least_squares_work_length(size_t m, size_t n)                                               // LS:642-646
{
    return mir_box_qp_work_length(n) + n * 5 + n * n + n * m + m * 2;
}
size_t mir_least_squares_iwork_length(size_t m, size_t n)                                              // LS:651-656
{
    (void)m;
    const size_t a = mir_box_qp_iwork_length(n);
    return a > n ? a : n;
}

size_t mir_box_qp_iwork_length_ilp64(size_t n) { return n + (n / sizeof(int64_t) + (n % sizeof(int64_t) != 0)); }   // QP:47-50, lapackint = long
size_t mir_least_squares_iwork_length_ilp64(size_t m, size_t n)
{
    (void)m;
    const size_t a = mir_box_qp_iwork_length_ilp64(n);
    return a > n ? a : n;
}

const char* mir_least_squares_status_string(mir_least_squares_status st)                               // LS:528-557, 666-669
{
    switch (st) {
    case mir_ls_furtherImprovement: return "The algorithm cann't improve the solution";
    case mir_ls_maxIterations: return "Maximum number of iterations reached";
    case mir_ls_xConverged: return "X converged";
    case mir_ls_gConverged: return "Jacobian converged";
    case mir_ls_fConverged: return "Residual is small enough";
    case mir_ls_badBounds: return "Initial guess must be within bounds.";
    case mir_ls_badGuess: return "Initial guess must be an array of finite numbers.";
    case mir_ls_badMinStepQuality: return "0 <= minStepQuality < 1 must hold.";
    case mir_ls_badGoodStepQuality: return "0 < goodStepQuality <= 1 must hold.";
    case mir_ls_badStepQuality: return "minStepQuality < goodStepQuality must hold.";
    case mir_ls_badLambdaParams: return "1 <= lambdaIncrease && lambdaIncrease <= T.max.sqrt and T.min_normal.sqrt <= lambdaDecrease && lambdaDecrease <= 1 must hold.";
    case mir_ls_numericError: return "Numeric Error";
    }
    return "";
}

void mir_least_squares_init_d(mir_least_squares_settings_d* s)                                         // LS:93-122, 761-764
{
    s->maxIterations = 1000; s->maxAge = 0;
    s->jacobianEpsilon = 0x1p-26;            // 2 ^^ ((1 - 53) / 2) (quirk Q10)
    s->absTolerance = DBL_EPSILON; s->relTolerance = 0; s->gradTolerance = DBL_EPSILON;
    s->maxGoodResidual = DBL_EPSILON * DBL_EPSILON;
    s->maxStep = std::sqrt(DBL_MAX) / 16; s->maxLambda = DBL_MAX / 16; s->minLambda = DBL_MIN * 16;
    s->minStepQuality = 0.1; s->goodStepQuality = 0.5; s->lambdaIncrease = 2;
    s->lambdaDecrease = (double)0.30901699437494742410229341718281905886L;   // 1 / (GoldenRatio * 2)
    s->qpSettings.relTolerance = DBL_EPSILON * 16; s->qpSettings.absTolerance = DBL_EPSILON * 16;
    s->qpSettings.maxIterations = 0;
}
void mir_least_squares_init_s(mir_least_squares_settings_s* s)                                         // LS:767-770
{
    s->maxIterations = 1000; s->maxAge = 0;
    s->jacobianEpsilon = 0x1p-11f;           // 2 ^^ ((1 - 24) / 2), integer division
    s->absTolerance = FLT_EPSILON; s->relTolerance = 0; s->gradTolerance = FLT_EPSILON;
    s->maxGoodResidual = FLT_EPSILON * FLT_EPSILON;
    s->maxStep = std::sqrt(FLT_MAX) / 16; s->maxLambda = FLT_MAX / 16; s->minLambda = FLT_MIN * 16;
    s->minStepQuality = 0.1f; s->goodStepQuality = 0.5f; s->lambdaIncrease = 2;
    s->lambdaDecrease = (float)0.30901699437494742410229341718281905886L;
    s->qpSettings.relTolerance = FLT_EPSILON * 16; s->qpSettings.absTolerance = FLT_EPSILON * 16;
    s->qpSettings.maxIterations = 0;
}
void mir_least_squares_reset_d(mir_least_squares_settings_d* s) { mir_least_squares_init_d(s); }        // LS:783-786
void mir_least_squares_reset_s(mir_least_squares_settings_s* s) { mir_least_squares_init_s(s); }        // LS:789-792

mir_least_squares_result_d mir_optimize_least_squares_d(                                               // LS:705-724
    const mir_least_squares_settings_d* settings, size_t m, size_t n, double* x, const double* l, const double* u,
    mir_slice_d work, mir_slice_i iwork, void* fContext, mir_least_squares_function_d f, void* gContext,
    mir_least_squares_jacobian_d g, void* tmContext, mir_least_squares_thread_manager tm)
{
    (void)work; (void)iwork;
    return solve_entry<double>(settings, m, n, x, l, u, nullptr, fContext, f, gContext, g, tmContext, tm);
}

mir_least_squares_result_s mir_optimize_least_squares_s(                                               // LS:729-748
    const mir_least_squares_settings_s* settings, size_t m, size_t n, float* x, const float* l, const float* u,
    mir_slice_s work, mir_slice_i iwork, void* fContext, mir_least_squares_function_s f, void* gContext,
    mir_least_squares_jacobian_s g, void* tmContext, mir_least_squares_thread_manager tm)
{
    (void)work; (void)iwork;
    return solve_entry<float>(settings, m, n, x, l, u, nullptr, fContext, f, gContext, g, tmContext, tm);
}

mir_least_squares_result_d mir_optimize_least_squares_gpu_d(
    const mir_least_squares_settings_d* settings, size_t m, size_t n, double* x, const double* l, const double* u,
    const mir_lsq_gpu_options* options, void* fContext, mir_least_squares_function_d f, void* gContext,
    mir_least_squares_jacobian_d g, void* tmContext, mir_least_squares_thread_manager tm)
{
    return solve_entry<double>(settings, m, n, x, l, u, options, fContext, f, gContext, g, tmContext, tm);
}

mir_least_squares_result_s mir_optimize_least_squares_gpu_s(
    const mir_least_squares_settings_s* settings, size_t m, size_t n, float* x, const float* l, const float* u,
    const mir_lsq_gpu_options* options, void* fContext, mir_least_squares_function_s f, void* gContext,
    mir_least_squares_jacobian_s g, void* tmContext, mir_least_squares_thread_manager tm)
{
    return solve_entry<float>(settings, m, n, x, l, u, options, fContext, f, gContext, g, tmContext, tm);
}

// ---- standalone BOXCQP ---------------------------------------------------------------------
}  // extern "C"

namespace {
template <typename T, typename QS>
int box_qp_entry(const QS* settings, size_t n_, const T* P, const T* q, const T* l, const T* u, T* x,
                 int unconstrainedSolution, int* iterations)
{
    if (iterations) *iterations = 0;
    if (n_ == 0) return mir_box_qp_solved;
    if (!device_available()) return mir_box_qp_numericError;
    const int n = (int)n_;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t oP = take(sizeof(T) * n * n), oq = take(sizeof(T) * n), ol = take(sizeof(T) * n), ou = take(sizeof(T) * n),
                 ox = take(sizeof(T) * n), oPm = take(sizeof(T) * n * n), oA = take(sizeof(T) * n * n),
                 oF = take(sizeof(T) * n * (n | 1)), ov = take(sizeof(T) * 12 * n), oi = take(sizeof(int32_t) * 2 * n),
                 oo = take(sizeof(int) * 4);
    char* base = nullptr;
    if (hipMalloc((void**)&base, off) != hipSuccess) return mir_box_qp_numericError;
    BoxQpArgs<T> a{};
    a.P = (T*)(base + oP); a.q = (T*)(base + oq); a.l = (T*)(base + ol); a.u = (T*)(base + ou); a.x = (T*)(base + ox);
    a.sc.Pm = (T*)(base + oPm); a.sc.A = (T*)(base + oA); a.sc.Fg = (T*)(base + oF); a.sc.vec = (T*)(base + ov);
    a.sc.ivec = (int32_t*)(base + oi); a.out = (int*)(base + oo); a.sc.dbg = nullptr;
    a.relTol = settings->relTolerance; a.absTol = settings->absTolerance; a.maxIterations = settings->maxIterations;
    a.unconstrained = unconstrainedSolution; a.n = n;
    const int nb = solve_nb(n, (int)sizeof(T));
    a.f_in_lds = nb > 0;
    const size_t lds = solve_lds_bytes(n, (int)sizeof(T));
    int out[2] = {mir_box_qp_numericError, 0};
    bool good = hipMemcpy((void*)a.P, P, sizeof(T) * n * n, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy((void*)a.q, q, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy((void*)a.l, l, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy((void*)a.u, u, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy((void*)a.x, x, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess;
    if (good && n > kSolveMaxN) {
        // any n: the 512-thread kernel of solve_big.h
        auto kern = k_box_qp_big<T>;
        good = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(BigLds<T>)) == hipSuccess;
        if (good) hipLaunchKernelGGL(kern, dim3(1), dim3(kBigThreads), sizeof(BigLds<T>), 0, a);
    } else if (good) {
        auto launch = [&](auto kern) {
            if (lds > 48 * 1024
                && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
                return false;
            hipLaunchKernelGGL(kern, dim3(1), dim3(kSolveThreads), lds, 0, a);
            return true;
        };
        switch (nb) {
        case 1: good = launch(k_box_qp<T, 1>); break;
        case 2: good = launch(k_box_qp<T, 2>); break;
        case 4: good = launch(k_box_qp<T, 4>); break;
        case 8: good = launch(k_box_qp<T, 8>); break;
        default: good = launch(k_box_qp<T, 0>); break;
        }
    }
    if (good) {
        good = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess
            && hipMemcpy(out, a.out, sizeof(out), hipMemcpyDeviceToHost) == hipSuccess
            && hipMemcpy(x, a.x, sizeof(T) * n, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(base);
    if (!good) return mir_box_qp_numericError;
    if (iterations) *iterations = out[1];
    return out[0];
}

template <typename T>
int jtj_entry(size_t m, size_t n, T* J, const T* y, const T* y_old, const T* dx, int broyden, T* JJ, T* Jy,
              void* stream_, float* kernel_ms, uint32_t variant = 0)
{
    if (!device_available()) return -1;
    if (n == 0 || m == 0) return -2;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const JtjPlan plan = jtj_plan<T>(m, (int)n, query_num_cu(), variant);
    const size_t packed_len = n * (n + 1) / 2 + n + 8;
    T *slabs = nullptr, *packed = nullptr, *dxdot = nullptr;
    LmState<T>* st = nullptr;
    size_t slab_count = (size_t)plan.nblk * plan.njobs * plan.slab_len;
    if ((size_t)plan.pc32_nblk * plan.slab_len > slab_count) slab_count = (size_t)plan.pc32_nblk * plan.slab_len;
    if (hipMalloc((void**)&slabs, sizeof(T) * slab_count) != hipSuccess) return -3;
    if (hipMalloc((void**)&packed, sizeof(T) * packed_len) != hipSuccess) { (void)hipFree(slabs); return -3; }
    if (hipMalloc((void**)&st, sizeof(LmState<T>) + sizeof(T) * 8) != hipSuccess) { (void)hipFree(slabs); (void)hipFree(packed); return -3; }
    dxdot = reinterpret_cast<T*>(st + 1);
    int rc = 0;
    if (broyden) {
        // ||dx||^2 on the device (n-vector, one block)
        hipLaunchKernelGGL(k_sumsq_partial<T>, dim3(1), dim3(256), 0, stream, dx, n, packed);
        hipLaunchKernelGGL(k_sumsq_final<T>, dim3(1), dim3(256), 0, stream, packed, 1, dxdot);
    }
    JtjArgs<T> a{};
    a.J = J; a.Jout = J; a.y = y; a.y_old = y_old; a.dx = dx; a.dx_dot = dxdot; a.slabs = slabs; a.m = m; a.n = (int)n;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, stream);
    if (jtj_run<T>(plan, a, broyden != 0, packed, stream, variant) != hipSuccess) rc = -4;
    (void)hipEventRecord(e1, stream);
    hipLaunchKernelGGL(k_unpack_grad<T>, dim3((unsigned)n + 1), dim3(128), 0, stream, packed, (int)n, JJ, Jy, st);
    if (hipStreamSynchronize(stream) != hipSuccess) rc = -5;
    if (kernel_ms) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); *kernel_ms = ms; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(slabs); (void)hipFree(packed); (void)hipFree(st);
    return rc;
}
}  // namespace

namespace {
// one wavefront = one workgroup per problem; every pointer in `a` is a device pointer. A model with a per-row
// basis (BatchedModel::nb > 0) gets its table from a stream-ordered allocation filled by k_batched_basis on the same stream.
template <int MODEL>
hipError_t batched_launch_model(BatchedArgs a, hipStream_t stream)
{
    constexpr int n = BatchedModel<MODEL>::n, nb = BatchedModel<MODEL>::nb;
    const size_t lds = (size_t)(n + 2) * a.m * sizeof(float);
    const unsigned blocks = (unsigned)a.count;
    auto kern = k_lm_batched<MODEL>;
    MIRLSQ_ENSURE_LDS(kern, lds);
    float* table = nullptr;
    bool pooled = true;                                          // stream-ordered allocation; plain hipMalloc where the runtime has no pools
    if (nb > 0) {
        const size_t rows = (size_t)(a.t_stride ? a.count : 1) * a.m;
        hipError_t e = hipMallocAsync((void**)&table, rows * nb * sizeof(float), stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            pooled = false;
            e = hipMalloc((void**)&table, rows * nb * sizeof(float));
            if (e != hipSuccess) return e;
        }
        const unsigned bb = (unsigned)std::min<size_t>((rows + 255) / 256, 4096);
        hipLaunchKernelGGL(k_batched_basis<MODEL>, dim3(bb), dim3(256), 0, stream, a.t, table, rows);
        a.basis = table;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (table) {
        hipError_t f;
        if (pooled) f = hipFreeAsync(table, stream);
        else { f = hipStreamSynchronize(stream); (void)hipFree(table); }      // the kernel reads the table: wait before freeing it
        if (e == hipSuccess) e = f;
    }
    return e;
}
}  // namespace

extern "C" {

// ---- batched one-wave-per-problem entry (cfg 5) ------------------------------------------------
namespace {
struct BatchedFallbackCtx { const float* t; const float* d; hipStream_t stream; int model; };
void batched_fallback_f(void* vctx, size_t m, size_t n, const float* x, float* y)
{
    (void)n;
    auto* c = static_cast<BatchedFallbackCtx*>(vctx);
    const unsigned blocks = (unsigned)((m + 255) / 256);
    if (c->model == kModelExpDecay)
        hipLaunchKernelGGL(k_batched_model_eval<kModelExpDecay>, dim3(blocks), dim3(256), 0, c->stream, c->t, c->d, x, y, (int)m);
    else if (c->model == kModelExp3Affine)
        hipLaunchKernelGGL(k_batched_model_eval<kModelExp3Affine>, dim3(blocks), dim3(256), 0, c->stream, c->t, c->d, x, y, (int)m);
    else
        hipLaunchKernelGGL(k_batched_model_eval<kModelExpDecayPad8>, dim3(blocks), dim3(256), 0, c->stream, c->t, c->d, x, y, (int)m);
}
inline int batched_model_n(int model)
{
    return model == kModelExpDecay ? 3 : ((model == kModelExp3Affine || model == kModelExpDecayPad8) ? 8 : 0);
}

// fill the settings part of the kernel arguments
void batched_settings(BatchedArgs& a, const mir_least_squares_settings_s* S)
{
    a.set.jacobianEpsilon = S->jacobianEpsilon; a.set.absTolerance = S->absTolerance; a.set.relTolerance = S->relTolerance;
    a.set.gradTolerance = S->gradTolerance; a.set.maxGoodResidual = S->maxGoodResidual; a.set.maxStep = S->maxStep;
    a.set.maxLambda = S->maxLambda; a.set.minLambda = S->minLambda; a.set.minStepQuality = S->minStepQuality;
    a.set.goodStepQuality = S->goodStepQuality; a.set.lambdaIncrease = S->lambdaIncrease; a.set.lambdaDecrease = S->lambdaDecrease;
    a.set.qpRelTolerance = S->qpSettings.relTolerance; a.set.qpAbsTolerance = S->qpSettings.absTolerance;
    a.set.qpMaxIterations = S->qpSettings.maxIterations;
    a.maxIterations = S->maxIterations; a.maxAge = S->maxAge;
}

hipError_t batched_launch(const BatchedArgs& a, int model, hipStream_t stream)
{
    if (model == kModelExpDecay) return batched_launch_model<kModelExpDecay>(a, stream);
    if (model == kModelExp3Affine) return batched_launch_model<kModelExp3Affine>(a, stream);
    return batched_launch_model<kModelExpDecayPad8>(a, stream);
}
}  // namespace

#ifdef MIRLSQ_BATCHED_TIMING
namespace { uint64_t* g_batched_timing = nullptr; size_t g_batched_timing_count = 0; }
// profiling builds: the cycle counters of the last mir_lsq_batched_kernel_s launch (6 per problem), after a synchronisation
int mir_lsq_batched_timing(uint64_t* host, size_t count)
{
    if (!g_batched_timing || count > g_batched_timing_count) return -1;
    return hipMemcpy(host, g_batched_timing, count * 10 * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
namespace { std::atomic<uint32_t> g_batched_variant{0}; }
void mir_lsq_batched_set_variant(uint32_t variant) { g_batched_variant.store(variant); }

int mir_lsq_batched_kernel_s(const mir_least_squares_settings_s* S, size_t count, size_t m, int model, float* x,
                             const float* lower, const float* upper, const float* t, size_t t_stride, const float* data,
                             mir_least_squares_result_s* results, void* stream)
{
    const int n = batched_model_n(model);
    if (!S || !x || !lower || !upper || !t || !data || !results || n == 0 || (t_stride != 0 && t_stride != m)) return -1;
    if (count == 0) return 0;
    if (!device_available()) return -2;
    if (m == 0 || (size_t)(n + 2) * m * sizeof(float) > 160 * 1024 - 512) return -3;
    static_assert(sizeof(BatchedResult) == sizeof(mir_least_squares_result_s), "the kernel writes the C result records in place");
    BatchedArgs a{};
    batched_settings(a, S);
    a.count = (int)count; a.m = (int)m; a.t_stride = (int)t_stride;
    a.variant = g_batched_variant.load();
#ifdef MIRLSQ_BATCHED_TIMING
    if (g_batched_timing_count < count) {
        if (g_batched_timing) (void)hipFree(g_batched_timing);
        g_batched_timing_count = 0;
        if (hipMalloc((void**)&g_batched_timing, count * 10 * sizeof(uint64_t)) != hipSuccess) return -4;
        g_batched_timing_count = count;
    }
    a.timing = g_batched_timing;
#endif
    a.t = t; a.data = data; a.x = x; a.lower = lower; a.upper = upper;
    a.results = reinterpret_cast<BatchedResult*>(results);
    return batched_launch(a, model, static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -5;
}

int mir_lsq_batched_posvx_s(size_t count, size_t n, const float* P, const float* rhs, float* x, int* info, void* stream)
{
    if (!P || !rhs || !x || !info || (n != 3 && n != 8)) return -1;
    if (count == 0) return 0;
    if (!device_available()) return -2;
    const unsigned blocks = (unsigned)std::min<size_t>(count, 8192);
    if (n == 8) hipLaunchKernelGGL(k_posvx_rows<8>, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), P, rhs, (int)count, x, info);
    else hipLaunchKernelGGL(k_posvx_rows<3>, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), P, rhs, (int)count, x, info);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int mir_optimize_least_squares_batched_s(const mir_least_squares_settings_s* S, size_t count, size_t m, int model,
                                         float* x, const float* lower, const float* upper,
                                         const float* t, size_t t_stride, const float* data,
                                         mir_least_squares_result_s* results)
{
    if (!S || !x || !lower || !upper || !t || !data || !results) return -1;
    const int n = batched_model_n(model);
    if (n == 0 || (t_stride != 0 && t_stride != m)) return -1;
    for (size_t i = 0; i < count; ++i) {       // defaults of LeastSquaresResult!T, LS:132-142
        results[i].status = mir_ls_numericError; results[i].iterations = results[i].fCalls = results[i].gCalls = 0;
        results[i].residual = Lim<float>::inf(); results[i].lambda = 0;
    }
    if (count == 0) return 0;
    // settings validation LS:934-943, common to all problems (codes reported per problem)
    int bad = 0;
    if (!(0 <= S->minStepQuality && S->minStepQuality < 1)) bad = mir_ls_badMinStepQuality;
    else if (!(0 <= S->goodStepQuality && S->goodStepQuality <= 1)) bad = mir_ls_badGoodStepQuality;
    else if (!(S->minStepQuality < S->goodStepQuality)) bad = mir_ls_badStepQuality;
    else if (!(1 <= S->lambdaIncrease && S->lambdaIncrease <= std::sqrt(FLT_MAX))) bad = mir_ls_badLambdaParams;
    else if (!(std::sqrt(FLT_MIN) <= S->lambdaDecrease && S->lambdaDecrease <= 1)) bad = mir_ls_badLambdaParams;
    if (!device_available()) return -2;
    const size_t lds = (size_t)(n + 2) * m * sizeof(float);
    if (m == 0 || lds > 160 * 1024 - 512) {
        std::fprintf(stderr, "[mir_optim_amd] batched entry: m = %zu does not fit one wave's LDS slice\n", m);
        return -3;
    }
    BatchedArgs a{};
    batched_settings(a, S);
    a.count = (int)count; a.m = (int)m;
    a.t_stride = (int)t_stride;
    a.variant = g_batched_variant.load();
    const size_t tb = (t_stride ? count : 1) * m * sizeof(float), db = count * m * sizeof(float), xb = count * n * sizeof(float);
    char* base = nullptr;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t ot = take(tb), od = take(db), ox = take(xb), ol = take(n * sizeof(float)), ou = take(n * sizeof(float)),
                 orr = take(count * sizeof(BatchedResult));
    if (hipMalloc((void**)&base, off) != hipSuccess) return -4;
    bool good = hipMemcpy(base + ot, t, tb, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(base + od, data, db, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(base + ox, x, xb, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(base + ol, lower, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(base + ou, upper, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    a.t = (const float*)(base + ot); a.data = (const float*)(base + od); a.x = (float*)(base + ox);
    a.lower = (const float*)(base + ol); a.upper = (const float*)(base + ou); a.results = (BatchedResult*)(base + orr);
    std::vector<BatchedResult> res(count);
    std::vector<float> x0(x, x + count * n);       // starts, for the fallback problems
    if (good && !bad) {
        good = batched_launch(a, model, nullptr) == hipSuccess;
        good = good && hipDeviceSynchronize() == hipSuccess
            && hipMemcpy(res.data(), a.results, count * sizeof(BatchedResult), hipMemcpyDeviceToHost) == hipSuccess
            && hipMemcpy(x, a.x, xb, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (good) {
        for (size_t i = 0; i < count; ++i) {
            if (bad) { results[i].status = bad; continue; }
            results[i].status = res[i].status; results[i].iterations = res[i].iterations; results[i].fCalls = res[i].fCalls;
            results[i].gCalls = res[i].gCalls; results[i].residual = res[i].residual; results[i].lambda = res[i].lambda;
            if (res[i].status == kBatchedNeedsGeneral) {
                // bounded step: complete this problem with the general solver (device callbacks, BOXCQP on the device)
                hipStream_t st = nullptr;
                if (hipStreamCreate(&st) != hipSuccess) { good = false; break; }
                BatchedFallbackCtx c{a.t + (t_stride ? i * m : 0), a.data + i * m, st, model};
                mir_lsq_gpu_options o{};
                o.struct_size = sizeof o; o.flags = MIR_LSQ_DEVICE_CALLBACKS; o.stream = st;
                std::memcpy(x + i * n, x0.data() + i * n, n * sizeof(float));
                results[i] = mir_optimize_least_squares_gpu_s(S, m, n, x + i * n, lower, upper, &o, &c, batched_fallback_f,
                                                              nullptr, nullptr, nullptr, nullptr);
                (void)hipStreamDestroy(st);
            }
        }
    }
    (void)hipFree(base);
    return good ? 0 : -5;
}

int mir_solve_box_qp_gpu_d(const mir_box_qp_settings_d* settings, size_t n, const double* P, const double* q,
                           const double* l, const double* u, double* x, int unconstrainedSolution, int* iterations)
{
    return box_qp_entry<double>(settings, n, P, q, l, u, x, unconstrainedSolution, iterations);
}
int mir_solve_box_qp_gpu_s(const mir_box_qp_settings_s* settings, size_t n, const float* P, const float* q,
                           const float* l, const float* u, float* x, int unconstrainedSolution, int* iterations)
{
    return box_qp_entry<float>(settings, n, P, q, l, u, x, unconstrainedSolution, iterations);
}

int mir_lsq_jtj_d(size_t m, size_t n, double* J, const double* y, const double* y_old, const double* dx, int broyden,
                  double* JJ, double* Jy, void* stream, float* kernel_ms)
{
    return jtj_entry<double>(m, n, J, y, y_old, dx, broyden, JJ, Jy, stream, kernel_ms);
}
int mir_lsq_jtj_variant_d(size_t m, size_t n, double* J, const double* y, const double* y_old, const double* dx, int broyden,
                          double* JJ, double* Jy, void* stream, float* kernel_ms, uint32_t variant)
{
    return jtj_entry<double>(m, n, J, y, y_old, dx, broyden, JJ, Jy, stream, kernel_ms, variant);
}
namespace {
int fd_jtj_entry(size_t m, size_t n, const double* Yrm, const double* twh, const double* y, double* J,
                 double* JJ, double* Jy, void* stream_, float* kernel_ms, bool diff);
}
int mir_lsq_fd_jtj_d(size_t m, size_t n, const double* Yrm, const double* twh, const double* y, double* J,
                     double* JJ, double* Jy, void* stream_, float* kernel_ms)
{
    return fd_jtj_entry(m, n, Yrm, twh, y, J, JJ, Jy, stream_, kernel_ms, false);
}
int mir_lsq_fd_diff_jtj_d(size_t m, size_t n, const double* Drm, const double* twh, const double* y, double* J,
                          double* JJ, double* Jy, void* stream_, float* kernel_ms)
{
    return fd_jtj_entry(m, n, Drm, twh, y, J, JJ, Jy, stream_, kernel_ms, true);
}
namespace {
int fd_jtj_entry(size_t m, size_t n, const double* Yrm, const double* twh, const double* y, double* J,
                 double* JJ, double* Jy, void* stream_, float* kernel_ms, bool diff)
{
    if (!device_available()) return -1;
    if (n == 0 || m == 0) return -2;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const JtjPlan plan = jtj_plan<double>(m, (int)n, query_num_cu());
    if (diff ? !jtj_fd_diff_ok(plan, (int)n) : (!plan.fdp && !plan.fdp8)) return -6;   // shape not covered by a fused kernel
    const size_t packed_len = n * (n + 1) / 2 + n + 8;
    double *slabs = nullptr, *packed = nullptr;
    LmState<double>* st = nullptr;
    const size_t slab_count = plan.fdp8 ? (size_t)plan.fdp8_nblk * plan.fdp8_slab_len : (size_t)plan.nblk * plan.njobs * plan.slab_len;
    if (hipMalloc((void**)&slabs, sizeof(double) * slab_count) != hipSuccess) return -3;
    if (hipMalloc((void**)&packed, sizeof(double) * packed_len) != hipSuccess) { (void)hipFree(slabs); return -3; }
    if (hipMalloc((void**)&st, sizeof(LmState<double>)) != hipSuccess) { (void)hipFree(slabs); (void)hipFree(packed); return -3; }
    int rc = 0;
    JtjArgs<double> a{};
    a.J = Yrm; a.Jout = J; a.y = y; a.y_old = y; a.dx = nullptr; a.dx_dot = nullptr; a.slabs = slabs; a.m = m; a.n = (int)n;
    a.twh = twh;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, stream);
    if ((diff ? jtj_run_fd_diff<double>(plan, a, packed, stream) : jtj_run_fd<double>(plan, a, packed, stream)) != hipSuccess) rc = -4;
    (void)hipEventRecord(e1, stream);
    hipLaunchKernelGGL(k_unpack_grad<double>, dim3((unsigned)n + 1), dim3(128), 0, stream, packed, (int)n, JJ, Jy, st);
    if (hipStreamSynchronize(stream) != hipSuccess) rc = -5;
    if (kernel_ms) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); *kernel_ms = ms; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(slabs); (void)hipFree(packed); (void)hipFree(st);
    return rc;
}
}  // namespace
int mir_lsq_jtj_s(size_t m, size_t n, float* J, const float* y, const float* y_old, const float* dx, int broyden,
                  float* JJ, float* Jy, void* stream, float* kernel_ms)
{
    return jtj_entry<float>(m, n, J, y, y_old, dx, broyden, JJ, Jy, stream, kernel_ms);
}

namespace {
__global__ void k_preload() {}
}
mir_lsq_workspace* mir_lsq_workspace_create(size_t m, size_t n, size_t elem_size)
{
    if (!device_available() || n == 0 || m == 0) return nullptr;
    // HIP loads a library's device code at its first kernel launch (tens of ms for this one): do it here, where the
    // caller sets things up, rather than in the first solve
    hipLaunchKernelGGL(k_preload, dim3(1), dim3(1), 0, nullptr);
    (void)hipStreamSynchronize(nullptr);
    if (elem_size == 8) return workspace_create<double>(m, n);
    if (elem_size == 4) return workspace_create<float>(m, n);
    return nullptr;
}
void mir_lsq_workspace_destroy(mir_lsq_workspace* ws) { workspace_destroy(ws); }

// ---- communicators ---------------------------------------------------------------------------
int mir_lsq_rccl_unique_id(void* out)
{
    void* h = rccl_open();
    if (!h) { std::fprintf(stderr, "[mir_optim_amd] librccl not found\n"); return -1; }
    auto fn = reinterpret_cast<int (*)(NcclUniqueId*)>(dlsym(h, "ncclGetUniqueId"));
    if (!fn) return -2;
    return fn(static_cast<NcclUniqueId*>(out));
}

mir_lsq_comm* mir_lsq_comm_create_rccl(int nranks, int rank, const void* unique_id)
{
    int preloaded = 0;
    void* h = rccl_open(&preloaded);
    if (!h) { std::fprintf(stderr, "[mir_optim_amd] librccl not found\n"); return nullptr; }
    auto init = reinterpret_cast<int (*)(void**, int, NcclUniqueId, int)>(dlsym(h, "ncclCommInitRank"));
    auto ar = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
    auto destroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    if (!init || !ar || !destroy) return nullptr;
    NcclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    void* c = nullptr;
    const int rc = init(&c, nranks, id, rank);
    if (rc != 0) { std::fprintf(stderr, "[mir_optim_amd] ncclCommInitRank failed: %d\n", rc); return nullptr; }
    auto* comm = new mir_lsq_comm();
    comm->nranks = nranks; comm->rank = rank; comm->kind = 1; comm->lib = h; comm->nccl_comm = c;
    comm->allreduce_fn = ar; comm->destroy_fn = destroy;
    comm->lib_preloaded = preloaded;
    {
        Dl_info info{};
        if (dladdr(reinterpret_cast<void*>(ar), &info) && info.dli_fname) std::snprintf(comm->lib_path, sizeof comm->lib_path, "%s", info.dli_fname);
        auto ver = reinterpret_cast<int (*)(int*)>(dlsym(h, "ncclGetVersion"));
        if (ver) (void)ver(&comm->lib_version);
        if (rank == 0)
            std::fprintf(stderr, "[mir_optim_amd] RCCL bound from %s (version %d, %s), %d ranks\n", comm->lib_path, comm->lib_version,
                         preloaded ? "already mapped by the host program" : "loaded by this library", nranks);
    }
    {
        // RCCL loads its kernels and connects its channels at the first collective of each size class: do that here
        // (creation is collective anyway), with the three payload sizes of a solve -- one scalar, the Broyden sweep
        // vector, the packed [J^T J | J^T y] -- so that the caller's first solve does not pay for it
        double* w = nullptr;
        const size_t sizes[3] = {1, 1024, 40000};
        if (hipMalloc((void**)&w, sizes[2] * sizeof(double)) == hipSuccess) {
            (void)hipMemset(w, 0, sizes[2] * sizeof(double));
            for (size_t sz : sizes)
                if (ar(w, w, sz, 8 /*ncclDouble*/, 0 /*ncclSum*/, c, nullptr) != 0) break;
            (void)hipStreamSynchronize(nullptr);
            (void)hipFree(w);
        }
    }
    return comm;
}

mir_lsq_comm* mir_lsq_comm_create_callback(int nranks, int rank, mir_lsq_allreduce_fn fn, void* ctx)
{
    if (!fn) return nullptr;
    auto* comm = new mir_lsq_comm();
    comm->nranks = nranks; comm->rank = rank; comm->kind = 2; comm->cb = fn; comm->cb_ctx = ctx;
    return comm;
}

int mir_lsq_comm_create_local_group(int nranks, mir_lsq_comm** out_comms)
{
    if (nranks < 1 || !out_comms) return -1;
    auto* g = new LocalGroup();
    g->nranks = nranks; g->refs = nranks;
    g->slots[0].resize(nranks); g->slots[1].resize(nranks); g->total.resize(nranks);
    for (int r = 0; r < nranks; ++r) {
        auto* c = new mir_lsq_comm();
        c->nranks = nranks; c->rank = r; c->kind = 3; c->group = g;
        out_comms[r] = c;
    }
    return 0;
}

int mir_lsq_comm_allreduce_d(mir_lsq_comm* comm, double* buf, size_t count, void* stream)
{
    return comm ? mirlsq::comm_allreduce<double>(comm, buf, count, static_cast<hipStream_t>(stream)) : -1;
}
int mir_lsq_comm_allreduce_s(mir_lsq_comm* comm, float* buf, size_t count, void* stream)
{
    return comm ? mirlsq::comm_allreduce<float>(comm, buf, count, static_cast<hipStream_t>(stream)) : -1;
}

int mir_lsq_comm_ranks(const mir_lsq_comm* comm)
{
    if (!comm) return -1;
    if (comm->kind == 1) {
        auto fn = reinterpret_cast<int (*)(void*, int*)>(dlsym(comm->lib, "ncclCommCount"));
        int cnt = -1;
        if (!fn || fn(comm->nccl_comm, &cnt) != 0) return -1;
        return cnt;
    }
    return comm->nranks;
}

int mir_lsq_comm_describe(const mir_lsq_comm* comm, char* buf, size_t len)
{
    if (!comm || !buf || len == 0) return -1;
    if (comm->kind == 1)
        return std::snprintf(buf, len, "rccl path=%s version=%d preloaded=%d ranks=%d rank=%d", comm->lib_path, comm->lib_version,
                             comm->lib_preloaded, mir_lsq_comm_ranks(comm), comm->rank);
    return std::snprintf(buf, len, "%s ranks=%d rank=%d", comm->kind == 2 ? "callback" : "local-group", comm->nranks, comm->rank);
}

void mir_lsq_comm_destroy(mir_lsq_comm* comm)
{
    if (!comm) return;
    if (comm->kind == 1 && comm->destroy_fn && comm->nccl_comm) comm->destroy_fn(comm->nccl_comm);
    if (comm->kind == 3 && comm->group) {
        // The handles of a group may be destroyed independently, each by its own rank's thread as soon as that rank is done:
        // a slower peer may still be summing this rank's slot of the last all-reduce (the sum runs outside the lock, after
        // the barrier), so every slot stays allocated until the LAST handle goes.
        LocalGroup* g = comm->group;
        bool last;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            last = --g->refs == 0;
        }
        if (last) {
            for (int par = 0; par < 2; ++par)
                for (auto& sl : g->slots[par]) if (sl.host) (void)hipHostFree(sl.host);
            for (auto& t : g->total) if (t.host) (void)hipHostFree(t.host);
            delete g;
        }
    }
    delete comm;
}

// ---- small device utilities ------------------------------------------------------------------
int mir_lsq_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}
void* mir_lsq_device_malloc(size_t bytes)
{
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
void mir_lsq_device_free(void* p) { if (p) (void)hipFree(p); }
int mir_lsq_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
int mir_lsq_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
int mir_lsq_selftest_reductions(int rounds, int mismatches[4])
{
    if (!mismatches || rounds <= 0) return -1;
    if (!device_available()) return -2;
    int* d = nullptr;
    if (hipMalloc((void**)&d, 4 * sizeof(int)) != hipSuccess) return -3;
    bool ok = hipMemset(d, 0, 4 * sizeof(int)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_selftest_reductions, dim3(2048), dim3(256), 0, nullptr, rounds, 12345u, d);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess
          && hipMemcpy(mismatches, d, 4 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    return ok ? 0 : -4;
}
int mir_lsq_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream)
{
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -1;
}
void* mir_lsq_stream_create(void)
{
    hipStream_t s = nullptr;
    if (hipStreamCreate(&s) != hipSuccess) return nullptr;
    return s;
}
void mir_lsq_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream)); }
int mir_lsq_stream_synchronize(void* stream) { return hipStreamSynchronize(static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -1; }
const char* mir_lsq_version(void) { return "mir_optim_amd 0.1 (gfx950)"; }

}  // extern "C"
