// workspace.hip -- device buffers of one (m, n, element type) problem (mir_lsq_workspace): one carve for everything the
// LM loop keeps in HBM + the m-sized side buffers; reusable across calls, a solve allocates nothing.
// Sizes follow the reference's own carve of `work` (least_squares.d:913-926) where a buffer has a counterpart there.
#include "driver.h"
#include "launch_util.h"

namespace mirlsq {

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

int query_num_cu()
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

template <typename T>
Buffers<T> carve(void* base, size_t m, size_t n, int num_cu)
{
    Buffers<T> b{};
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
    b.slabs = (T*)take(jtj_slab_elems(jtj_plan<T>(m, (int)n, num_cu)), sizeof(T));
    b.lrD = (T*)take((size_t)kLrMax * n, sizeof(T));
    b.lrvec = (T*)take((size_t)lr_len((int)n) + 6, sizeof(T));
    b.lrpart = (T*)take((size_t)lr_blocks(m, num_cu) * lr_len((int)n), sizeof(T));
    for (int k = 0; k < kChainMax; ++k) {
        b.sc[k].Pm = (T*)take(n * n, sizeof(T));
        b.sc[k].A = (T*)take(n * n, sizeof(T));
        b.sc[k].Fg = (T*)take(n * (n | 1), sizeof(T));
        b.sc[k].vec = (T*)take(12 * n, sizeof(T));
        b.sc[k].ivec = (int32_t*)take(2 * n, sizeof(int32_t));
        b.sc[k].dbg = nullptr;
        b.sc[k].coop = (unsigned long long*)take(n > (size_t)kSolveMaxN ? kCoopWords : 0, sizeof(unsigned long long));
        b.sc[k].cS = (T*)take(coop_scratch_elems((int)n), sizeof(T));
    }
    b.sc[0].dbg = (long long*)take(32, sizeof(long long));
    b.bytes = off;
    return b;
}

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
    // the helpers' sync words start from zero ONCE: every later value carries the number of its launch (solve_coop.h)
    if (n > (size_t)kSolveMaxN) {
        const Buffers<T> bb = carve<T>(ws->dev, m, n, ws->num_cu);
        for (int k = 0; k < kChainMax; ++k)
            if (hipMemset(bb.sc[k].coop, 0, kCoopWords * sizeof(unsigned long long)) != hipSuccess) {
                std::fprintf(stderr, "[mir_optim_amd] workspace: hipMemset failed\n");
                workspace_destroy(ws);
                return nullptr;
            }
        (void)hipStreamSynchronize(nullptr);                      // (the memsets ran on the null stream)
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
    for (int k = 0; k < mir_lsq_workspace::kCopyStreams; ++k) {
        if (ws->copy_event[k]) (void)hipEventDestroy(ws->copy_event[k]);
        if (ws->copy_stream[k]) (void)hipStreamDestroy(ws->copy_stream[k]);
    }
    delete ws;
}

template Buffers<double> carve<double>(void*, size_t, size_t, int);
template Buffers<float> carve<float>(void*, size_t, size_t, int);
template mir_lsq_workspace* workspace_create<double>(size_t, size_t);
template mir_lsq_workspace* workspace_create<float>(size_t, size_t);

}  // namespace mirlsq


extern "C" {

mir_lsq_workspace* mir_lsq_workspace_create(size_t m, size_t n, size_t elem_size)
{
    if (!mirlsq::device_available() || n == 0 || m == 0) return nullptr;
    // HIP loads a library's device code at its first kernel launch (tens of ms for this one): do it here, where the
    // caller sets things up, rather than in the first solve
    mirlsq::preload_jtj(); mirlsq::preload_broyden(); mirlsq::preload_loop(); mirlsq::preload_jacobian();
    if (elem_size == 8) mirlsq::preload_solve_d(); else mirlsq::preload_solve_s();
    (void)hipStreamSynchronize(nullptr);
    if (elem_size == 8) return mirlsq::workspace_create<double>(m, n);
    if (elem_size == 4) return mirlsq::workspace_create<float>(m, n);
    return nullptr;
}
void mir_lsq_workspace_destroy(mir_lsq_workspace* ws) { mirlsq::workspace_destroy(ws); }

}  // extern "C"
