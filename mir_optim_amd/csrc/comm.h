// comm.h -- row-shard communicator. The reference has no communication
// layer; this is the one exchange step the MI355X design adds: a sum all-reduce of the packed
// [J^T J lower | J^T y] buffer per Jacobian-changing pass and of one scalar per trial step.
// RCCL is loaded with dlopen so the library also loads on hosts without it.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/mir_optim_amd.h"

namespace mirlsq {
// Shared state of an in-process group (mir_lsq_comm_create_local_group): per-rank pinned host slots, double-buffered by
// call parity so that one barrier per all-reduce suffices (a rank can be at most one call ahead of the slowest one).
struct LocalGroup {
    int nranks = 0;
    int refs = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;                       // a rank timed out or failed: every later wait returns at once
    struct Slot { void* host = nullptr; size_t bytes = 0; };
    std::vector<Slot> slots[2];                // [parity][rank]: the rank's contribution
    std::vector<Slot> total;                   // [rank]: the sum, staged for the copy back
    // returns false on timeout / broken group
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == nranks) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        const bool ok = cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; });
        if (!ok || broken) { broken = true; cv.notify_all(); return false; }
        return true;
    }
};
}  // namespace mirlsq

struct mir_lsq_comm {
    int nranks = 1, rank = 0;
    int kind = 0;   // 1 = rccl, 2 = callback, 3 = in-process group, 4 = replay of recorded totals
    mirlsq::LocalGroup* group = nullptr;
    uint64_t local_calls = 0;
    // rccl
    void* lib = nullptr;
    void* nccl_comm = nullptr;
    char lib_path[256] = {0};     // the shared object ncclAllReduce was bound from (dladdr)
    int lib_version = 0;          // ncclGetVersion: major * 10000 + minor * 100 + patch
    int lib_preloaded = 0;        // 1: the host program had an RCCL mapped already (e.g. PyTorch's) and that one was bound
    int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*destroy_fn)(void*) = nullptr;
    // callback
    mir_lsq_allreduce_fn cb = nullptr;
    void* cb_ctx = nullptr;
    double* cb_scratch = nullptr;   // device doubles for float problems
    size_t cb_scratch_len = 0;
    // record (kind 3): the totals of this rank's all-reduces, appended to a caller-owned host buffer (mir_lsq_comm_record)
    double* rec_buf = nullptr;
    size_t rec_cap = 0, rec_len = 0;
    bool rec_overflow = false;
    // replay (kind 4): recorded totals on the device, a cursor, optionally an inner communicator every exchange also passes through
    double* replay_dev = nullptr;
    size_t replay_len = 0, replay_pos = 0;
    mir_lsq_comm* replay_inner = nullptr;
    uint32_t replay_delay_us = 0;   // MODEL of a collective's latency: every replayed exchange first holds the stream this long
};

namespace mirlsq {

struct NcclUniqueId { char internal[128]; };

inline void* rccl_open(int* preloaded = nullptr)
{
    // prefer an RCCL that the host program has already loaded (e.g. the one bundled with PyTorch): two
    // RCCL instances in one process work but double the IPC/bootstrap state. Which one was bound, and its version, is
    // recorded in the communicator (mir_lsq_comm_describe) so that a version skew shows up in the first log line.
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (preloaded) *preloaded = h != nullptr;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return h;
}

// comm.hip: one wave that leaves after `us` microseconds (the replay communicator's latency model)
void comm_replay_delay(uint32_t us, hipStream_t stream);

// sum `count` elements of device buffer `buf` over all ranks, in place, ordered on `stream`
template <typename T>
inline int comm_allreduce(mir_lsq_comm* c, T* buf, size_t count, hipStream_t stream)
{
    if (!c) return 0;
    // a communicator with one rank still goes through its collective (RCCL: a copy on the stream): the world-1 tests and
    // bench runs exercise exactly the calls an N-rank job makes
    if (c->kind == 1) {
        const int dtype = sizeof(T) == 8 ? 8 /*ncclDouble*/ : 7 /*ncclFloat*/;
        const int rc = c->allreduce_fn(buf, buf, count, dtype, 0 /*ncclSum*/, c->nccl_comm, stream);
        if (rc != 0) { std::fprintf(stderr, "[mir_optim_amd] ncclAllReduce failed: %d\n", rc); return -1; }
        return 0;
    }
    if (c->kind == 3) {
        LocalGroup* g = c->group;
        const size_t bytes = count * sizeof(T);
        const int par = (int)(c->local_calls++ & 1);
        auto fit = [&](LocalGroup::Slot& sl) {
            if (sl.bytes >= bytes) return true;
            if (sl.host) (void)hipHostFree(sl.host);
            sl.host = nullptr; sl.bytes = 0;
            size_t want = bytes < 4096 ? 4096 : bytes;
            if (hipHostMalloc(&sl.host, want, hipHostMallocDefault) != hipSuccess) return false;
            sl.bytes = want;
            return true;
        };
        LocalGroup::Slot& mine = g->slots[par][c->rank];
        LocalGroup::Slot& tot = g->total[c->rank];
        bool ok = fit(mine) && fit(tot)
            && hipMemcpyAsync(mine.host, buf, bytes, hipMemcpyDeviceToHost, stream) == hipSuccess
            && hipStreamSynchronize(stream) == hipSuccess;
        if (!ok) { std::lock_guard<std::mutex> lk(g->mu); g->broken = true; g->cv.notify_all(); }
        if (!g->barrier() || !ok) { std::fprintf(stderr, "[mir_optim_amd] in-process all-reduce failed (rank %d)\n", c->rank); return -1; }
        T* out = static_cast<T*>(tot.host);
        const T* first = static_cast<const T*>(g->slots[par][0].host);
        for (size_t i = 0; i < count; ++i) out[i] = first[i];
        for (int r = 1; r < g->nranks; ++r) {                 // rank order: the same bits on every rank
            const T* src = static_cast<const T*>(g->slots[par][r].host);
            for (size_t i = 0; i < count; ++i) out[i] += src[i];
        }
        if (c->rec_buf) {
            if (c->rec_len + count <= c->rec_cap) { for (size_t i = 0; i < count; ++i) c->rec_buf[c->rec_len + i] = (double)out[i]; c->rec_len += count; }
            else c->rec_overflow = true;
        }
        if (hipMemcpyAsync(buf, out, bytes, hipMemcpyHostToDevice, stream) != hipSuccess
            || hipStreamSynchronize(stream) != hipSuccess) return -1;
        return 0;
    }
    if (c->kind == 4) {
        // replay: the recorded total of this exchange replaces the buffer (stream-ordered copy, no host synchronisation)
        if constexpr (sizeof(T) == 8) {
            if (c->replay_pos + count > c->replay_len) {
                std::fprintf(stderr, "[mir_optim_amd] replay communicator: the tape ends at %zu, exchange needs %zu more\n", c->replay_len, count);
                return -1;
            }
            if (c->replay_delay_us) comm_replay_delay(c->replay_delay_us, stream);
            if (hipMemcpyAsync(buf, c->replay_dev + c->replay_pos, count * sizeof(double), hipMemcpyDeviceToDevice, stream) != hipSuccess) return -1;
            c->replay_pos += count;
            return c->replay_inner ? comm_allreduce<T>(c->replay_inner, buf, count, stream) : 0;
        } else {
            std::fprintf(stderr, "[mir_optim_amd] replay communicator supports f64 problems only\n");
            return -1;
        }
    }
    if (c->kind == 2) {
        if constexpr (sizeof(T) == 8) {
            c->cb(c->cb_ctx, reinterpret_cast<double*>(buf), count, stream);
            return 0;
        } else {
            std::fprintf(stderr, "[mir_optim_amd] callback communicator supports f64 problems only\n");
            return -1;
        }
    }
    return -1;
}

}  // namespace mirlsq
