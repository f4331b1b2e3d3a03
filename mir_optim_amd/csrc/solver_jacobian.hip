// solver_jacobian.hip -- the Jacobian side of an LM pass (/root/reference/source/mir/optim/least_squares.d, LS:nnn):
//   Broyden passes as read-only sweeps with pending rank-one terms (LS:999-1007 -> broyden_lr.h), flush + resynchronisation;
//   full refreshes: finite differences through device callbacks (LS:1016-1050 restructured into panels), through host
//   callbacks with the reference's thread-manager contract (LS:1018-1049 literally), or the analytic g (LS:1011-1015);
//   J^T y and J^T J (LS:1052, 1065) with the row-shard all-reduce behind them.
#include "driver.h"
#include "launch_util.h"
#include "misc_kernels.h"

namespace mirlsq {

// ---- Broyden pass without rewriting J (LS:1003-1006, 1052, 1065): sweep, reduce, all-reduce, n x n finish
template <typename T>
bool Solver<T>::broyden_lowrank(const T* y_dev, const T* yold_dev)
{
    T* U = static_cast<T*>(ws->ulr);
    if (lr_k >= lr_cap) {
        const uint64_t launches_before = launches_now();
        struct Excl { Solver* s; uint64_t l0; ~Excl() { s->launches_excluded += s->launches_now() - l0; } } excl{this, launches_before};
        // fold the pending rank-one terms into J (the reference's successive `ger`s, LS:1006) ...
        if (!ok(lr_flush<T>(B.J, U, B.lrD, lr_k, m, (int)n, ws->num_cu, stream), "broyden flush")) return false;
        lr_k = 0;
        if (stats) stats->broyden_flushes++;
        // ... and resynchronise: J^T J and J^T y_old recomputed from the flushed J, as the reference's syrk / gemv do
        // every pass (LS:1052, 1065). The recurrence J^T J += v dx^T + dx v^T + uu dx dx^T below then never runs for more
        // than lr_cap passes on its own rounding errors (ill-conditioned problems: cancellation could otherwise
        // accumulate over up to maxAge = 2n passes). y_old: the sweep expects J_{k-1}^T J_{k-1} in JJ; Jy is
        // rebuilt by every sweep anyway.
        if (!plain_products(yold_dev)) return false;
        if (stats) stats->jtj_resyncs++;
    }
    // (the pass of a fused round is enqueued by enqueue_fused_tail, solver_loop.hip: the same kernels, the sweep ahead of the
    // decision and the n x n side inside the solve's launch)
    LrArgs<T> a{};
    a.J = B.J; a.U = U; a.D = B.lrD; a.dx = B.dx_acc; a.dx_dot = &B.st->dx_dot; a.y = y_dev; a.y_old = yold_dev;
    a.partials = B.lrpart; a.m = m; a.n = (int)n; a.k = lr_k;
    const int nblk = lr_blocks(m, ws->num_cu), len = lr_len((int)n);
    ev_begin(1);
    if (!ok(lr_sweep<T>(a, nblk, stream), "broyden sweep")) return false;
    ev_end();
    if (!ok(lr_reduce<T>(B.lrpart, nblk, (int)n, B.lrvec, stream), "broyden reduce")) return false;
    if (comm && !allreduce(B.lrvec, (size_t)len, 1)) return false;
    if (!ok(lr_finish<T>(B.lrvec, B.lrD, B.dx_acc, lr_k, (int)n, B.JJ, B.Jy, B.st, stream), "broyden finish")) return false;
    if (stats) stats->broyden_lr_columns += (uint64_t)lr_k;
    ++lr_k;
    return true;
}

// J^T J, J^T y_vec of the J in memory -> JJ, Jy (all-reduced, unpacked)
template <typename T>
bool Solver<T>::plain_products(const T* y_vec)
{
    JtjArgs<T> a{};
    a.J = B.J; a.Jout = B.J; a.y = y_vec; a.y_old = y_vec; a.dx = B.dx_acc; a.dx_dot = &B.st->dx_dot;
    a.slabs = B.slabs; a.m = m; a.n = (int)n;
    const bool direct = unpack_in_reduce(false);
    ev_begin(0);
    if (!ok(jtj_run<T>(plan, a, false, B.packed, stream, direct ? unpack_target() : JtjUnpack<T>{}), "jtj kernel")) return false;
    ev_end();
    return finish_products(direct);
}

// Single GPU: the slab reduction writes J^T J (both triangles) and J^T y itself (the solve kernel takes |J^T y|_inf); with a
// communicator the packed buffer is all-reduced first and k_unpack_grad expands it
template <typename T>
bool Solver<T>::finish_products(bool direct)
{
    if (direct) return ok(hipGetLastError(), "jtj reduce");
    if (comm && !allreduce(B.packed, (size_t)n * (n + 1) / 2 + n, 0)) return false;
    return ok(jtj_unpack<T>(B.packed, (int)n, B.JJ, B.Jy, B.st, stream), "unpack");
}

template <typename T>
bool Solver<T>::jacobian_products(bool broyden, const T* y_dev, const T* yold_dev)
{
    if (broyden && lowrank) return broyden_lowrank(y_dev, yold_dev);
    if (!broyden) lr_k = 0;                             // J was refreshed in full: nothing is pending any more
    JtjArgs<T> a{};
    a.J = B.J; a.Jout = B.J; a.y = y_dev; a.y_old = yold_dev; a.dx = B.dx_acc; a.dx_dot = &B.st->dx_dot;
    a.slabs = B.slabs; a.m = m; a.n = (int)n;
    if (!broyden && fd_fused) {
        // the row-major FD panel is still in ws->ypanel: one kernel forms the Jacobian rows (LS:1041-1047), writes
        // them to J and accumulates J^T J / J^T y from the same registers
        const bool diff = fd_fused == 2;
        fd_fused = 0;
        a.J = static_cast<const T*>(ws->ypanel); a.twh = B.twh;
        const bool direct = unpack_in_reduce(true);
        const JtjUnpack<T> u = direct ? unpack_target() : JtjUnpack<T>{};
        ev_begin(3);
        if (!ok(diff ? jtj_run_fd_diff<T>(plan, a, B.packed, stream, u) : jtj_run_fd<T>(plan, a, B.packed, stream, u), "fd + jtj kernel")) return false;
        ev_end();
        return finish_products(direct);
    }
    const bool direct = unpack_in_reduce(false, broyden);
    ev_begin(broyden ? 1 : 0);
    if (!ok(jtj_run<T>(plan, a, broyden, B.packed, stream, direct ? unpack_target() : JtjUnpack<T>{}), "jtj kernel")) return false;
    ev_end();
    return finish_products(direct);
}

// ---- finite-difference Jacobian, device callbacks (LS:1016-1050 restructured: all perturbed
//      points are generated at once, evaluated one by one or in batches, and written to J in
//      coalesced column panels)
template <typename T>
bool Solver<T>::fd_device()
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
        // all 2n points in one sweep, Y[i][2j], Y[i][2j+1] = f(x + h e_j)_i, f(x - h e_j)_i; k_jtj_fdp<., true>
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
template <typename T>
void Solver<T>::fd_task_trampoline(mir_least_squares_task task, uint32_t totalThreads, uint32_t threadId, uint32_t j)
{
    static_cast<Solver<T>*>(task.context)->fd_task(totalThreads, threadId, j);
}
template <typename T>
void Solver<T>::fd_task(uint32_t totalThreads, uint32_t threadId, uint32_t j)
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
    // (a worker thread of the manager: its launch goes into the solve's own counter, not into this thread's tl_launches)
    hipLaunchKernelGGL(k_fd_fill_col<T>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, mB, fr, twh, B.J, m, (int)n, (int)j);
    worker_launches.fetch_add(1, std::memory_order_relaxed);
    if (hipStreamSynchronize(stream) != hipSuccess) fd_failed = true;
}
// Can this refresh stage through the pinned point-major panel? Needs the whole 2n x m panel on both sides: device (the
// budget fd_device() uses: half of the free HBM, at most 64 GiB) and pinned host memory (at most 16 GiB). Otherwise the
// column-at-a-time path below (per-slot staging vectors, one strided column write per task) serves any size.
template <typename T>
bool Solver<T>::fd_host_prepare_panel()
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
template <typename T>
bool Solver<T>::fd_host()
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

template <typename T>
bool Solver<T>::analytic_jacobian()
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

#define MIRLSQ_INSTANTIATE(T)                                                                       \
    template bool Solver<T>::broyden_lowrank(const T*, const T*);                                   \
    template bool Solver<T>::plain_products(const T*);                                              \
    template bool Solver<T>::finish_products(bool);                                                 \
    template bool Solver<T>::jacobian_products(bool, const T*, const T*);                           \
    template bool Solver<T>::fd_device();                                                           \
    template void Solver<T>::fd_task_trampoline(mir_least_squares_task, uint32_t, uint32_t, uint32_t); \
    template void Solver<T>::fd_task(uint32_t, uint32_t, uint32_t);                                 \
    template bool Solver<T>::fd_host_prepare_panel();                                               \
    template bool Solver<T>::fd_host();                                                             \
    template bool Solver<T>::analytic_jacobian();
MIRLSQ_INSTANTIATE(double)
MIRLSQ_INSTANTIATE(float)
#undef MIRLSQ_INSTANTIATE

}  // namespace mirlsq

MIRLSQ_DEFINE_PRELOAD(jacobian)
