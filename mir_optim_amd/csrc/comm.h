// comm.h -- row-shard communicator. The reference has no communication
// layer; this is the one exchange step the MI355X design adds: a sum all-reduce of the packed
// [J^T J lower | J^T y] buffer per Jacobian-changing pass and of one scalar per trial step.
// RCCL is loaded with dlopen so the library also loads on hosts without it.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/mir_optim_amd.h"

struct mir_lsq_comm {
    int nranks = 1, rank = 0;
    int kind = 0;   // 1 = rccl, 2 = callback
    // rccl
    void* lib = nullptr;
    void* nccl_comm = nullptr;
    int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*destroy_fn)(void*) = nullptr;
    // callback
    mir_lsq_allreduce_fn cb = nullptr;
    void* cb_ctx = nullptr;
    double* cb_scratch = nullptr;   // device doubles for float problems
    size_t cb_scratch_len = 0;
};

namespace mirlsq {

struct NcclUniqueId { char internal[128]; };

inline void* rccl_open()
{
    // prefer an RCCL that the host program has already loaded (e.g. the one bundled with PyTorch): two
    // RCCL instances in one process work but double the IPC/bootstrap state
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return h;
}

// sum `count` elements of device buffer `buf` over all ranks, in place, ordered on `stream`
template <typename T>
inline int comm_allreduce(mir_lsq_comm* c, T* buf, size_t count, hipStream_t stream)
{
    if (!c || c->nranks <= 1) return 0;
    if (c->kind == 1) {
        const int dtype = sizeof(T) == 8 ? 8 /*ncclDouble*/ : 7 /*ncclFloat*/;
        const int rc = c->allreduce_fn(buf, buf, count, dtype, 0 /*ncclSum*/, c->nccl_comm, stream);
        if (rc != 0) { std::fprintf(stderr, "[mir_optim_amd] ncclAllReduce failed: %d\n", rc); return -1; }
        return 0;
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
