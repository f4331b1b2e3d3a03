// malloc_async_probe.hip -- does ROCm's stream-ordered memory pool hand out a block that overlaps a live allocation?
//
// Round 4's batched launch took its per-launch basis table from hipMallocAsync / hipFreeAsync, and about one launch in 150 with
// a 2 MB table returned wrong fits for a contiguous range of problems; with the table in ordinary memory: 0 of 400
// (include/mir_optim_amd_batched.hpp, tests/test_gpu_batched.py::test_repeated_launches_with_a_large_basis_table_agree). The
// allocator was switched without knowing whether the library's ordering or the pool was at fault. This program has NO library
// code: a loop of
//     hipMallocAsync(table) -> fill kernel (pattern keyed by the iteration) -> consumer kernel (checks every word)
//     -> hipFreeAsync(table)
// on a stream, interleaved with hipMalloc / hipFree of 1 ... 64 MB buffers that are filled with another pattern by kernels on
// a SECOND stream (so a block that overlaps one of them is seen as corruption of either side), with and without
// hipDeviceSynchronize between iterations, on the null stream and on a created one, and in the library's exact call pattern
// (mode 2: the consumer runs for ~0.6 ms like k_lm_batched, hipFreeAsync is enqueued right behind it, and the next iteration's
// hipMallocAsync follows immediately on the same stream).
//
//   hipcc --offload-arch=gfx950 -O2 -o malloc_async_probe scripts/probes/malloc_async_probe.hip && ./malloc_async_probe [iters]
// Prints one line per configuration: iterations, mismatching words seen by the consumer, mismatching words in the bystander
// buffers, distinct device addresses the pool handed out.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorName(e_)); std::exit(2); } } while (0)

__device__ __host__ inline uint32_t pat(uint32_t key, uint32_t i) { uint32_t v = key * 2654435761u + i * 40503u; v ^= v >> 15; return v * 2246822519u; }

__global__ void k_fill(uint32_t* p, size_t words, uint32_t key)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = pat(key, (uint32_t)i);
}
// reads every word `passes` times (a long-running consumer re-reads its table, as the batched fit does at every residual
// evaluation) and counts words that are not the pattern
static int g_flags = 0;     // argv[2]: 1 = no bystander allocations, 2 = pool release threshold = max (the pool keeps its memory over
                            // synchronisations), 4 = plain instead of non-temporal loads in the consumer
__global__ void k_check(const uint32_t* p, size_t words, uint32_t key, int passes, unsigned long long* bad, int plain)
{
    unsigned long long b = 0;
    for (int s = 0; s < passes; ++s)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
            b += (plain ? ((const volatile uint32_t*)p)[i] : __builtin_nontemporal_load(p + i)) != pat(key, (uint32_t)i);
    if (b) atomicAdd(bad, b);
}

struct Result { long iters; unsigned long long bad_consumer, bad_bystander; size_t distinct; };

// mode 0: synchronise the device after every iteration; 1: never synchronise inside the loop; 2: the library's pattern (long consumer)
static Result run(long iters, bool null_stream, int mode, size_t table_bytes)
{
    hipStream_t s = nullptr, side = nullptr;
    if (!null_stream) CK(hipStreamCreate(&s));
    CK(hipStreamCreate(&side));
    unsigned long long *bad = nullptr, *bad2 = nullptr;
    CK(hipMalloc((void**)&bad, 16));
    CK(hipMemset(bad, 0, 16));
    bad2 = bad + 1;
    const size_t words = table_bytes / 4;
    std::set<void*> seen;
    struct By { uint32_t* p; size_t words; uint32_t key; };
    std::vector<By> live;
    uint32_t rng = 12345;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    for (long it = 0; it < iters; ++it) {
        uint32_t* t = nullptr;
        CK(hipMallocAsync((void**)&t, table_bytes, s));
        seen.insert(t);
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, s, t, words, (uint32_t)it);
        // a bystander comes or goes on the side stream: ordinary allocations of 1 .. 64 MB holding their own pattern
        if (g_flags & 1) {
        } else if ((next() & 3) != 0 && live.size() < 6) {
            By b; b.words = ((size_t)1 << (18 + next() % 7)); b.key = 0x80000000u + (uint32_t)it;
            CK(hipMalloc((void**)&b.p, b.words * 4));
            hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, side, b.p, b.words, b.key);
            live.push_back(b);
        } else if (!live.empty()) {
            const size_t k = next() % live.size();
            hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, side, live[k].p, live[k].words, live[k].key, 1, bad2, g_flags & 4);
            CK(hipStreamSynchronize(side));
            CK(hipFree(live[k].p));
            live.erase(live.begin() + (long)k);
        }
        hipLaunchKernelGGL(k_check, dim3(mode == 2 ? 2048 : 256), dim3(256), 0, s, t, words, (uint32_t)it, mode == 2 ? 40 : 1, bad, g_flags & 4);
        CK(hipFreeAsync(t, s));
        if (mode == 0) CK(hipDeviceSynchronize());
        if (mode == 2 && (it & 15) == 15) CK(hipStreamSynchronize(s));      // the host entry waits for results now and then
    }
    CK(hipDeviceSynchronize());
    for (auto& b : live) {
        hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, side, b.p, b.words, b.key, 1, bad2, g_flags & 4);
        CK(hipStreamSynchronize(side));
        CK(hipFree(b.p));
    }
    unsigned long long h[2] = {0, 0};
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    CK(hipFree(bad));
    if (s) CK(hipStreamDestroy(s));
    CK(hipStreamDestroy(side));
    return Result{iters, h[0], h[1], seen.size()};
}

// mode 3: the host entry's exact sequence in round 4 (batched.hip + launch_batched): hipMalloc(inputs) -> blocking H2D copies ->
// hipMallocAsync(table, null stream) -> table kernel -> long consumer reading BOTH -> hipFreeAsync -> hipDeviceSynchronize ->
// D2H copy -> hipFree(inputs); count = 256 problems of m = 512: 1 MB of inputs, a 2 MB table.
static Result run_library_sequence(long iters)
{
    const size_t in_bytes = 256 * 512 * 4 * 2 + 256 * 8 * 4 + 256 * 24 + 1024, table_bytes = (size_t)256 * 512 * 4 * 4;
    const size_t in_words = in_bytes / 4, t_words = table_bytes / 4;
    std::vector<uint32_t> host(in_words), back(64);
    unsigned long long* bad = nullptr;
    CK(hipMalloc((void**)&bad, 16));
    CK(hipMemset(bad, 0, 16));
    std::set<void*> seen;
    for (long it = 0; it < iters; ++it) {
        uint32_t* base = nullptr;
        CK(hipMalloc((void**)&base, in_bytes));
        for (size_t i = 0; i < in_words; ++i) host[i] = pat(0x40000000u + (uint32_t)it, (uint32_t)i);
        CK(hipMemcpy(base, host.data(), in_bytes, hipMemcpyHostToDevice));
        uint32_t* t = nullptr;
        CK(hipMallocAsync((void**)&t, table_bytes, nullptr));
        seen.insert(t);
        hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, nullptr, t, t_words, (uint32_t)it);
        hipLaunchKernelGGL(k_check, dim3(256), dim3(64), 0, nullptr, t, t_words, (uint32_t)it, 30, bad, g_flags & 4);
        hipLaunchKernelGGL(k_check, dim3(256), dim3(64), 0, nullptr, base, in_words, 0x40000000u + (uint32_t)it, 30, bad + 1, g_flags & 4);
        CK(hipFreeAsync(t, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(back.data(), base, 256, hipMemcpyDeviceToHost));
        CK(hipFree(base));
    }
    unsigned long long h[2] = {0, 0};
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    CK(hipFree(bad));
    return Result{iters, h[0], h[1], seen.size()};
}

int main(int argc, char** argv)
{
    const long iters = argc > 1 ? std::atol(argv[1]) : 2000;
    g_flags = argc > 2 ? std::atoi(argv[2]) : 0;
    if (g_flags & 2) {
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = UINT64_MAX;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    }
    int rt = 0;
    CK(hipRuntimeGetVersion(&rt));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("device %s  HIP runtime %d  iterations per configuration %ld  flags %d (1 no bystanders, 2 release threshold max, 4 plain loads)\n",
                prop.gcnArchName, rt, iters, g_flags);
    unsigned long long total = 0;
    for (int mode = 0; mode < 3; ++mode)
        for (int ns = 0; ns < 2; ++ns)
            for (size_t bytes : {(size_t)2 << 20, (size_t)8 << 20}) {
                const Result r = run(iters, ns == 1, mode, bytes);
                std::printf("mode %d (%s)  %s stream  table %zu MB: %ld iterations, consumer mismatches %llu, bystander mismatches %llu, "
                            "%zu distinct pool addresses\n", mode,
                            mode == 0 ? "device sync every iteration" : mode == 1 ? "no sync in the loop" : "library pattern: long consumer, free right behind",
                            ns ? "null" : "created", bytes >> 20, r.iters, r.bad_consumer, r.bad_bystander, r.distinct);
                std::fflush(stdout);
                total += r.bad_consumer + r.bad_bystander;
            }
    {
        const Result r = run_library_sequence(3 * iters);
        std::printf("mode 3 (the round-4 host entry's sequence: hipMalloc inputs, pool table, long consumers, free + device sync): %ld iterations, "
                    "table mismatches %llu, input mismatches %llu, %zu distinct pool addresses\n", r.iters, r.bad_consumer, r.bad_bystander, r.distinct);
        total += r.bad_consumer + r.bad_bystander;
    }
    std::printf("%s\n", total ? "CORRUPTION SEEN: the stream-ordered pool (or its ordering) reproduces without any library code"
                              : "no corruption in any configuration");
    return total ? 1 : 0;
}
