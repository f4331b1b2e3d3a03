// comm.hip -- row-shard communicators (the reference has no communication layer): creation,
// description and destruction of the RCCL (dlopen), callback, in-process-group and replay communicators of comm.h. The
// exchanges they carry are the three reductions of a pass, least_squares.d:1052, 1065, 1115.
#include "driver.h"

using namespace mirlsq;

namespace mirlsq {
__global__ void k_comm_replay_delay(long long ticks)          // 10 ns ticks; bounded: the wave leaves when the clock says so
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void comm_replay_delay(uint32_t us, hipStream_t stream)
{
    hipLaunchKernelGGL(k_comm_replay_delay, dim3(1), dim3(64), 0, stream, (long long)us * 100);
}
}  // namespace mirlsq

extern "C" {

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
    if (comm->kind == 4) {
        char inner[256] = "none";
        if (comm->replay_inner) (void)mir_lsq_comm_describe(comm->replay_inner, inner, sizeof inner);
        return std::snprintf(buf, len, "replay ranks=%d rank=%d tape=%zu doubles inner=[%s]", comm->nranks, comm->rank, comm->replay_len, inner);
    }
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
    if (comm->kind == 4 && comm->replay_dev) (void)hipFree(comm->replay_dev);
    delete comm;
}

int mir_lsq_comm_record(mir_lsq_comm* comm, double* host_buf, size_t capacity)
{
    if (!comm || comm->kind != 3) return -1;
    comm->rec_buf = host_buf; comm->rec_cap = host_buf ? capacity : 0; comm->rec_len = 0; comm->rec_overflow = false;
    return 0;
}
size_t mir_lsq_comm_recorded(const mir_lsq_comm* comm)
{
    if (!comm || comm->rec_overflow) return (size_t)-1;
    return comm->rec_len;
}
mir_lsq_comm* mir_lsq_comm_create_replay(int nranks, int rank, const double* totals_host, size_t len, mir_lsq_comm* inner)
{
    if (!totals_host || len == 0 || nranks < 1 || !device_available()) return nullptr;
    auto* comm = new mir_lsq_comm();
    comm->nranks = nranks; comm->rank = rank; comm->kind = 4; comm->replay_len = len; comm->replay_inner = inner;
    if (hipMalloc((void**)&comm->replay_dev, len * sizeof(double)) != hipSuccess
        || hipMemcpy(comm->replay_dev, totals_host, len * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        if (comm->replay_dev) (void)hipFree(comm->replay_dev);
        delete comm;
        return nullptr;
    }
    return comm;
}
int mir_lsq_comm_replay_set_delay(mir_lsq_comm* comm, unsigned microseconds)
{
    if (!comm || comm->kind != 4) return -1;
    comm->replay_delay_us = microseconds;
    return 0;
}
int mir_lsq_comm_replay_rewind(mir_lsq_comm* comm)
{
    if (!comm || comm->kind != 4) return -1;
    comm->replay_pos = 0;
    return 0;
}

}  // extern "C"
