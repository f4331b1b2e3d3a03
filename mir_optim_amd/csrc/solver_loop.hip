// solver_loop.hip -- Solver<T>::run(): the LM loop of optimizeLeastSquaresImplGeneric!T
// (/root/reference/source/mir/optim/least_squares.d:930-1176, LS:nnn) as rounds of kernels on one HIP stream: the n x n
// solve for a ladder of lambdas, the trial residuals, the decision (k_decide_chain), the host's wait for the decision
// point in the pinned mirror. The Jacobian side of a pass lives in solver_jacobian.hip; see driver.h.
#include "driver.h"
#include "launch_util.h"
#include "misc_kernels.h"

namespace mirlsq {

template <typename T>
void Solver<T>::ev_begin(int kind)
{
    if (!time_kernels) return;
    HpScope hp(this, 0);
    EventPair p{};
    p.kind = spec_enqueue ? kind + 100 : kind;       // the pass a fused round runs ahead: counted only once it is committed
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
template <typename T>
void Solver<T>::ev_end()
{
    if (!time_kernels) return;
    HpScope hp(this, 0);
    (void)hipEventRecord(events.back().b, stream);
}

template <typename T>
bool Solver<T>::setup()
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
    plan = jtj_plan<T>(m, (int)n, ws->num_cu);
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
    host_profile = (variant & MIR_LSQ_VARIANT_HOST_PROFILE) != 0;
    {
        const int v = (int)((variant >> MIR_LSQ_VARIANT_LR_CAP_SHIFT) & 31u);
        if (v >= 1 && v <= kLrMax) lr_cap = v;
    }
    y = B.y;
    mB = B.mB;
    fr = B.ytmp;
    big_solve = n > (uint32_t)kSolveMaxN || (variant & MIR_LSQ_VARIANT_SOLVE_GENERIC) != 0;
    if (n > (uint32_t)kLrMaxN) lowrank = false;    // the read-only Broyden sweep keeps n <= kLrMaxN = 512; above, J is rewritten
    // The FUSED round (enqueue_fused_tail): behind a round's one trial residual the Broyden sweep of the NEXT pass is run
    // speculatively (it needs the step and the two residual vectors, not the decision) and carries the trial's sum of squares,
    // ONE all-reduce serves LS:1115 and LS:1052 / 1065, and ONE kernel decides the trial, applies the pass's n x n side and
    // solves the next system: 3 launches and 1 collective per accepted Broyden pass instead of 7 and 2. Device callbacks, the
    // read-only sweep (n <= kLrMaxN) and the one-workgroup solves (n <= kSolveMaxN) only; a trace keeps
    // the one-by-one rounds (MIR_LSQ_VARIANT_NO_PIPELINE selects them too): the same bits either way.
    fused = device_cb && lowrank && !big_solve && !trace && !(variant & MIR_LSQ_VARIANT_NO_PIPELINE);
    twh_h.resize(n);
    // x, lower, upper sit back to back in the workspace: one copy from the pinned block instead of three from pageable
    // memory (each of those is a staged blit kernel, ~18 us apart on the stream)
    char* stage = static_cast<char*>(ws->pinned) + align_up(2 * sizeof(LmState<T>) + (3 * (size_t)n + 8) * sizeof(T), 256);
    const size_t ol = (size_t)(reinterpret_cast<char*>(B.lower) - reinterpret_cast<char*>(B.x));
    const size_t ou = (size_t)(reinterpret_cast<char*>(B.upper) - reinterpret_cast<char*>(B.x));
    std::memcpy(stage, xh, n * sizeof(T));
    std::memcpy(stage + ol, lh, n * sizeof(T));
    std::memcpy(stage + ou, uh, n * sizeof(T));
    return ok(hipMemcpyAsync(B.x, stage, ou + n * sizeof(T), hipMemcpyHostToDevice, stream), "H2D x, l, u");
}

template <typename T>
void Solver<T>::teardown()
{
    if (stream) (void)hipStreamSynchronize(stream);
    if (host_profile) std::fprintf(stderr, "[host profile] J %p  FD panel %p  y %p\n", (void*)B.J, ws ? ws->ypanel : nullptr, (void*)B.y);
    if (host_profile)
        std::fprintf(stderr, "[host profile] events %.3f ms  all-reduce calls %.3f  trial callbacks %.3f  readback+sync %.3f  solve launch %.3f  "
                             "longest single call %.3f\n", hp_ms[0], hp_ms[1], hp_ms[2], hp_ms[3], hp_ms[4], hp_ms[5]);
    if (stats) {
        for (auto& e : events) {
            if (e.kind < 0 || e.kind >= 100) continue;       // a fused round's pass run ahead that was not the next pass
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
template <typename T>
bool Solver<T>::eval_f(const T* x_dev, const T* x_host, T* y_dev)
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

// ---- ||v||^2 -> B.sum[slot] on device (all-reduced over row shards): the residual at entry, LS:955
template <typename T>
int Solver<T>::sumsq_blocks() const
{
    int nb = (int)((m + 4095) / 4096);
    if (nb > kPartials) nb = kPartials;
    return nb < 1 ? 1 : nb;
}
template <typename T>
bool Solver<T>::sumsq(const T* v, int slot)
{
    const int nb = sumsq_blocks();
    MIRLSQ_LAUNCH(k_sumsq_partial<T>, dim3(nb), dim3(256), 0, stream, v, m, B.partials, (size_t)0, kPartials);
    MIRLSQ_LAUNCH(k_sumsq_final<T>, dim3(1), dim3(256), 0, stream, B.partials, nb, B.sum + slot, kPartials);
    if (comm && !allreduce(B.sum + slot, 1, 2)) return false;
    return ok(hipGetLastError(), "sumsq");
}

// ---- ||v_k||^2 of the ks TRIAL residual vectors -> B.sum[1 + k]: the row walk and summation order of the Broyden sweep
//      (k_lr_sumsq), so that a trial's sum has the same bits whether it rode on a speculative sweep (fused round) or not.
//      Single GPU: stage 2 is left to k_decide_chain (sums_pending = the partial count).
template <typename T>
bool Solver<T>::trial_sums(const T* v, int count, size_t vstride)
{
    static_assert(kPartials >= kLrMaxBlocks, "one partial per workgroup of the sweep's grid");
    const int nb = lr_blocks(m, ws->num_cu);
    if (!ok(lr_sumsq<T>(v, m, count, vstride, B.partials, kPartials, nb, stream), "trial sums")) return false;
    if (!comm) { sums_pending = nb; return true; }
    if (!ok(lr_sumsq_final<T>(B.partials, kPartials, nb, count, B.sum + 1, stream), "trial sums, stage 2")) return false;
    return allreduce(B.sum + 1, (size_t)count, 2);
}

// ---- row-shard exchange: sum `count` elements over the ranks, in place, ordered on the stream. kind: 0 packed
//      [J^T J | J^T y], 1 Broyden sweep vector, 2 residual sums (mir_lsq_stats.allreduce_*): the three reductions of a pass,
//      LS:1052, 1065, 1115
template <typename T>
bool Solver<T>::allreduce(T* buf, size_t count, int kind)
{
    HpScope hp(this, 1);
    if (stats && !spec_enqueue) { stats->allreduce_calls[kind]++; stats->allreduce_elems[kind] += count; }
    return comm_allreduce<T>(comm, buf, count, stream) == 0;
}

// ---- optional per-pass trace (mir_lsq_trace)
template <typename T>
void Solver<T>::trace_emit(int event, uint32_t iterations, T lambda, T residual, T trial_residual, T dx_dot)
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
template <typename T>
bool Solver<T>::trace_round(int ks, T residual_before, uint32_t iterations_before)
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
template <typename T>
bool Solver<T>::wait_state(uint32_t expect)
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

template <typename T>
bool Solver<T>::read_state(const T* vec_dev)
{
    HpScope hp(this, 3);
    if (!ok(hipMemcpyAsync(st_h, B.st, sizeof(LmState<T>), hipMemcpyDeviceToHost, stream), "D2H state")) return false;
    if (vec_dev && !ok(hipMemcpyAsync(trial_h, vec_dev, n * sizeof(T), hipMemcpyDeviceToHost, stream), "D2H vec")) return false;
    return ok(hipStreamSynchronize(stream), "sync");
}

// MIR_LSQ_VARIANT_DEBUG_SOLVE: the phase stamps of the solve kernel that has just been enqueued (synchronises the stream)
template <typename T>
void Solver<T>::print_solve_dbg(bool fused_round)
{
    long long h[32];
    if (hipMemcpy(h, B.sc[0].dbg, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (fused_round)
        std::fprintf(stderr, "[solve dbg] fused round head (10ns ticks): decision + publication %lld (loads landed %lld, decided + point copied %lld, fence + sequence number %lld)  n x n side of the Broyden pass %lld\n", h[27] - h[26], h[29] - h[26], h[30] - h[29], h[31] - h[30], h[28] - h[27]);
    std::fprintf(stderr, "[solve dbg] (10ns ticks) build %lld  copy/equil %lld  scale %lld  potrf %lld  potrs %lld  refine %lld  epilogue %lld  total %lld  shader MHz %.0f  [matvec1 %lld berr %lld]\n",
                 h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[8] - h[7], h[8] - h[0], (double)(h[10] - h[9]) / (double)(h[8] - h[0]) * 100.0, h[11] - h[5], h[12] - h[11]);
    if (n <= 128)
        std::fprintf(stderr, "[solve dbg] load (10ns ticks): kernel entry -> loads landed %lld (of which inside posvx %lld)  reduction + LDS commit %lld  rest of the phase %lld\n",
                     h[24] - h[0], h[24] - h[2], h[25] - h[24], h[3] - h[25]);
    if (n <= 128)
        std::fprintf(stderr, "[solve dbg] refinement (10ns ticks): residual 1 %lld  berr 1 %lld  potrs 2 %lld  residual 2 %lld  rest %lld  (correction applied: %lld)\n",
                     h[11] - h[5], h[12] - h[11], h[13] - h[12], h[14] - h[13], h[6] - h[14], h[15]);
    if (n <= 128)
        std::fprintf(stderr, "[solve dbg] lds_potrf, wave 0 (shader cycles, summed over the panels): diagonal update + factor %lld  wait %lld  rows below %lld  wait %lld\n", h[20], h[21], h[22], h[23]);
    if (n > 128)
        std::fprintf(stderr, "[solve dbg] potrf_panel steps (10ns ticks, summed over the panels): earlier panels on MFMA %lld (of which publishing / waiting for the helpers' look-ahead jobs %lld)  diagonal rows %lld  other rows + store %lld\n", h[16], h[19], h[17], h[18]);
}

// the n x n part of a round (LS:1053-1110, 1141-1142) for ks ladder entries
template <typename T>
LmSolveArgs<T> Solver<T>::solve_args(int ks, const T* lam, bool check_grad, bool lambda_from_state)
{
    LmSolveArgs<T> a{};
    a.JJ = B.JJ; a.Jy = B.Jy; a.x = B.x; a.lower = B.lower; a.upper = B.upper;
    a.dx = B.dx; a.trial = B.trial; a.st = B.st; a.rec = B.rec; a.set = sd; a.n = (int)n;
    for (int k = 0; k < kChainMax; ++k) { a.sc[k] = B.sc[k]; a.lam[k] = (lam && k < ks) ? lam[k] : T(0); }
    a.f_in_lds = solve_nb((int)n, (int)sizeof(T)) > 0;
    a.check_grad = check_grad ? 1 : 0;
    a.lambda_from_state = lambda_from_state ? 1 : 0;
    if (!dbg_solve) a.sc[0].dbg = nullptr;
    // n > 256: every ladder entry's workgroup gets helpers (solve_coop.h)
    a.coop_w = (big_solve && n > (uint32_t)kSolveMaxN && !(variant & MIR_LSQ_VARIANT_SOLVE_ONE_WORKGROUP) && !coop_disabled) ? coop_peers((int)n) : 1;
    // an entry's workgroups hold a CU each (150 KB of LDS) and wait for one another: never more of them than half the device
    // (a partitioned or masked GPU), or a group could never be resident at once
    if (a.coop_w > ws->num_cu / 2) a.coop_w = ws->num_cu / 2 >= 2 ? ws->num_cu / 2 : 1;
    a.coop_epoch = a.coop_w > 1 ? ++ws->solve_epoch : 0;
    a.coop_absent = (variant & MIR_LSQ_VARIANT_DEBUG_HELPERS_ABSENT) ? 1 : 0;
    return a;
}
template <typename T>
bool Solver<T>::enqueue_solve(int ks, const T* lam, bool check_grad, bool lambda_from_state)
{
    const LmSolveArgs<T> a = solve_args(ks, lam, check_grad, lambda_from_state);
    ev_begin(2);
    {
        HpScope hp(this, 4);
        // all bounds infinite: the variant without the BOXCQP active-set loop (solve_kernel.h, BOUNDED = false)
        if (!ok(launch_lm_solve<T>(a, ks, has_bounds, big_solve, stream), "solve launch")) return false;
    }
    ev_end();
    return true;
}

// the decision of a round (LS:1080-1161) for ks trials; publishes decision point ++seq
template <typename T>
DecideArgs<T> Solver<T>::decide_args(int ks, bool check_grad, bool lambda_from_state)
{
    DecideArgs<T> d{};
    d.sums = B.sum + 1; d.rec = B.rec; d.st = B.st; d.set = sd; d.x = B.x; d.trial = B.trial; d.dx_chain = B.dx;
    d.dx_acc = B.dx_acc; d.n = (int)n; d.ks = ks; d.check_grad = check_grad ? 1 : 0;
    d.lambda_from_state = lambda_from_state ? 1 : 0;
    ++seq;
    d.host_st = st_slot_d[seq & 1]; d.host_x = x_slot_d[seq & 1]; d.seq = seq;
    d.maxIterations = S->maxIterations;
    d.partials = B.partials; d.nparts = sums_pending; d.pstride = kPartials;
    sums_pending = 0;
    return d;
}
//      sum_v != nullptr: the ks trial residual vectors at sum_v + k m still have to be summed (trial_sums() runs first; on a
//      single GPU its second stage is left to the decision kernel); else their sums of squares are in B.sum + 1
template <typename T>
bool Solver<T>::enqueue_decide(int ks, bool check_grad, bool lambda_from_state, const T* sum_v)
{
    if (sum_v && !trial_sums(sum_v, ks, m)) return false;
    const DecideArgs<T> d = decide_args(ks, check_grad, lambda_from_state);
    MIRLSQ_LAUNCH(k_decide_chain<T>, dim3(1), dim3(kSolveThreads), 0, stream, d);
    return ok(hipGetLastError(), "decide kernel");
}

// The tail of a FUSED round (see setup()), behind the round's one trial residual ytr = f(trial):
//   k_broyden_lr   SPECULATIVELY, as if the trial were accepted: y_new = ytr, y_old = y, the step and its dx.dx from ladder
//                  entry 0 (what the decision would copy into dx_acc / the state) -- plus ||ytr||^2 (entry lr_yy); when the
//                  entry's record already rules a Broyden pass out the kernel only forms that sum
//   k_lr_reduce    -> B.lrvec;  ONE all-reduce of [sweep | sum of squares] (LS:1115 and LS:1052 / 1065 of the next pass)
//   k_lm_solve     with a.fused: decision of this round (published at once), the pass's n x n side, the next pass's solve
// A rejected trial discards the sweep (column lr_k of U is rewritten by the next one). Host-side bookkeeping of the pass that
// may have been run ahead waits for commit_spec_round().
template <typename T>
bool Solver<T>::enqueue_fused_tail(const T* ytr, bool check_grad, bool lambda_from_state)
{
    spec_events_from = events.size();
    if (stats) stats->fused_rounds++;
    LrArgs<T> a{};
    a.J = B.J; a.U = static_cast<T*>(ws->ulr); a.D = B.lrD; a.dx = B.dx; a.dx_dot = &B.rec[0].new_dx_dot; a.y = ytr; a.y_old = y;
    a.partials = B.lrpart; a.m = m; a.n = (int)n; a.k = lr_k;
    a.spec_rec = B.rec; a.absTolerance = sd.absTolerance; a.relTolerance = sd.relTolerance;
    const int nblk = lr_blocks(m, ws->num_cu), len = lr_len((int)n);
    spec_enqueue = true;                                 // (the events of the pass run ahead count once it is committed)
    ev_begin(1);
    const bool swept = ok(lr_sweep<T>(a, nblk, stream), "broyden sweep");
    ev_end();
    spec_enqueue = false;
    if (!swept || !ok(lr_reduce<T>(B.lrpart, nblk, (int)n, B.lrvec, stream), "broyden reduce")) return false;
    if (comm && !allreduce(B.lrvec, (size_t)len, 1)) return false;
    LmSolveArgs<T> sa = solve_args(1, nullptr, true, false);
    sa.lambda_from_device = 1;
    sa.fused = 1;
    sa.dec = decide_args(1, check_grad, lambda_from_state);
    sa.dec.sums = B.lrvec + lr_yy((int)n);
    sa.dec.spec_static = 1;
    sa.fin_lr = B.lrvec; sa.fin_D = B.lrD; sa.fin_JJ = B.JJ; sa.fin_Jy = B.Jy; sa.fin_k = lr_k;
    spec_enqueue = true;
    ev_begin(2);
    bool good;
    {
        HpScope hp(this, 4);
        good = ok(launch_lm_solve<T>(sa, 1, has_bounds, false, stream), "fused round launch");
    }
    ev_end();
    spec_enqueue = false;
    return good;
}
// the pass run ahead by the fused round is the one the reference runs next: do now what the host does when it enqueues a
// Broyden round itself
template <typename T>
void Solver<T>::commit_spec_round()
{
    for (size_t i = spec_events_from; i < events.size(); ++i) if (events[i].kind >= 100) events[i].kind -= 100;
    if (stats) {
        stats->jacobian_broyden++;
        stats->fused_passes++;
        stats->broyden_lr_columns += (uint64_t)lr_k;
    }
    ++lr_k;
}
template <typename T>
void Solver<T>::drop_spec_round()
{
    for (size_t i = spec_events_from; i < events.size(); ++i) if (events[i].kind >= 100) events[i].kind = -1;
}

template <typename T>
typename Solver<T>::Result Solver<T>::run()
{
    ret.status = mir_ls_numericError; ret.iterations = 0; ret.fCalls = 0; ret.gCalls = 0;   // LS:132-142
    ret.residual = Lim<T>::inf(); ret.lambda = 0;
    const auto t_start = std::chrono::steady_clock::now();
    launches_mark = launches_now();

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
        if (!sumsq(y, 0)) { fail = true; break; }                        // LS:955
        MIRLSQ_LAUNCH(k_init_state<T>, dim3(1), dim3(1), 0, stream, B.sum, B.st, st_slot_d[seq & 1], seq);
        if (!ok(hipGetLastError(), "init state") || !wait_state(seq)) { fail = true; break; }
    } while (false);
    if (fail) { teardown(); ret.status = mir_ls_numericError; return ret; }

    ret.residual = st_h->residual;
    bool fConverged = ret.residual <= S->maxGoodResidual;                // LS:956
    bool needJacobian = true;                                            // LS:959
    bool last_rejected = false;
    bool spec_live = false;            // the Jacobian side and the solve of the round at the top of the loop have run already (fused round)
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
            // the Jacobian side and the solve of this round have run in the previous round's fused kernel
            spec_live = false;
            if (!needJacobian || !(age < maxAge) || lr_k >= lr_cap) {
                std::fprintf(stderr, "[mir_optim_amd] internal error: the pass run ahead by the fused round is not the next pass\n");
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
        if (dbg_solve && !solve_enqueued) print_solve_dbg(false);

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

        // Fused round: one trial now, and -- if it is accepted and no exit test fires (decided on the device) -- a Broyden pass
        // next that needs neither a full refresh (age) nor a flush of the pending terms (lr_k). One-bit predictor: only while
        // first trials are being accepted (the rejection tail of a noisy fit would waste a sweep per miss).
        const bool fuse = fused && spec_predict && ks == 1 && !skip_eval && age < maxAge && lr_k < lr_cap;
        if (fuse ? !enqueue_fused_tail(ytr, newJacobian, lambda_from_state)
                 : !enqueue_decide(ks, newJacobian, lambda_from_state, skip_eval ? nullptr : ytr)) { fail = true; break; }
        const uint32_t round_seq = seq;
        if (!wait_state(round_seq)) { fail = true; break; }
        if (fuse) {
            if (st_h->spec_ok) spec_live = true; else drop_spec_round();
            if (dbg_solve && spec_live) print_solve_dbg(true);
        }
        if (ks == 1 && newJacobian) spec_predict = st_h->decision == kDecideAccept;

        if (trace && !trace_round(ks, ret.residual, ret.iterations)) { fail = true; break; }
        if (st_h->coop_rescued) {
            // helper workgroups of the any-n solve did not answer within kCoopSpinSeconds (a GPU shared with other work, a
            // starved queue): the rescue launch has solved those entries on one workgroup; the rest of THIS solve does not ask
            // for helpers again (one stall per solve at most) and the caller's statistics say so
            if (stats) stats->coop_timeouts += st_h->coop_rescued;
            coop_disabled = true;
        }
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
        if (s.trace) s.trace->count = 0;
    }
    const typename Abi<T>::Result r = s.run();
    if (s.stats_user) std::memcpy(s.stats_user, &s.stats_local, s.stats_bytes);
    return r;
}

#define MIRLSQ_INSTANTIATE(T)                                                                                                   \
    template struct Solver<T>;                                                                                                  \
    template Abi<T>::Result solve_entry<T>(const Abi<T>::Settings*, size_t, size_t, T*, const T*, const T*, const mir_lsq_gpu_options*, \
                                           void*, Abi<T>::F, void*, Abi<T>::G, void*, mir_least_squares_thread_manager);
MIRLSQ_INSTANTIATE(double)
MIRLSQ_INSTANTIATE(float)
#undef MIRLSQ_INSTANTIATE

}  // namespace mirlsq

MIRLSQ_DEFINE_PRELOAD(loop)
