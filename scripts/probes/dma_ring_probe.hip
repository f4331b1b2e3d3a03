// Probe: how fast can a 4-wave workgroup stream a row-major array through an LDS-DMA ring?
// Variants isolate the cost of the barrier, the LDS reads and the MFMA chain.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_ring_probe scripts/probes/dma_ring_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;
using Acc = __attribute__((ext_vector_type(4))) double;

template <int D, int MODE, int NMFMA, int G = 1>   // MODE bit0: barrier, bit1: ds_read; G = 4 KB groups per stage
__global__ __launch_bounds__(256, 2) void k(const double* __restrict__ src, double* __restrict__ out, size_t nstages)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = D + 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per = (nstages + gridDim.x - 1) / gridDim.x;
    const size_t s0 = blockIdx.x * per;
    const size_t s1 = s0 + per < nstages ? s0 + per : nstages;
    const size_t S = s1 > s0 ? s1 - s0 : 0;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(src);
    auto issue = [&](size_t s) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t off = ((s0 + s) * G + g) * 4096 + (size_t)(wave * 64 + lane) * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr)(base + off), (lds_ptr)(smem + ((s % NS) * G + g) * 4096 + wave * 1024), 16, 0, 0);
        }
    };
    Acc acc[9];
    for (int i = 0; i < 9; ++i) acc[i] = Acc{0, 0, 0, 0};
    double sum = 0;
    const size_t pre = S < (size_t)D ? S : (size_t)D;
    for (size_t s = 0; s < pre; ++s) issue(s);
    for (size_t s = 0; s < S; ++s) {
        if (s + D < S) { issue(s + D); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * G) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MODE & 1) __builtin_amdgcn_s_barrier();
        if constexpr (MODE & 2) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
            const double* slot = reinterpret_cast<const double*>(smem + ((s % NS) * G + g) * 4096);
            double v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = slot[(lane >> 4) * 128 + 16 * c + (lane & 15)];
            if constexpr (NMFMA > 0) {
#pragma unroll
                for (int i = 0; i < NMFMA; ++i) acc[i % 9] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[i % 8], v[(i * 3) % 8], acc[i % 9], 0, 0, 0);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) sum += v[c];
            }
            }
        }
    }
    for (int i = 0; i < 9; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (sum == 12345.678) out[threadIdx.x] = sum;
}

template <int D, int MODE, int NMFMA, int G = 1>
void run(const double* src, double* out, size_t nstages4k, int grid, const char* name)
{
    auto kern = k<D, MODE, NMFMA, G>;
    const size_t nstages = nstages4k / G;
    const size_t lds = (D + 2) * 4096 * G;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, out, nstages);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r > 0 && ms < best) best = ms;
    }
    printf("%-44s grid %4d D %2d G %d: %.3f ms  %.0f GB/s\n", name, grid, D, G, best, nstages * G * 4096.0 / best / 1e6);
}

int main()
{
    const size_t nstages = 250000;   // 1.024 GB
    double *src, *out;
    hipMalloc(&src, nstages * 4096); hipMalloc(&out, 4096);
    hipMemset(src, 0, nstages * 4096);
    for (int grid : {256, 512, 768}) {
        run<14, 0, 0>(src, out, nstages, grid, "dma only (no barrier, no read)");
        run<14, 1, 0>(src, out, nstages, grid, "dma + barrier");
        run<14, 3, 0>(src, out, nstages, grid, "dma + barrier + ds_read");
        run<14, 3, 9>(src, out, nstages, grid, "dma + barrier + ds_read + 9 mfma");
        run<7, 3, 9>(src, out, nstages, grid, "same, D = 7");
        run<14, 3, 36>(src, out, nstages, grid, "dma + barrier + ds_read + 36 mfma");
        run<6, 3, 9, 2>(src, out, nstages, grid, "8 KB stages, 9 mfma per 4 KB");
        run<3, 3, 9, 4>(src, out, nstages, grid, "16 KB stages, 9 mfma per 4 KB");
        run<2, 3, 9, 4>(src, out, nstages, grid, "16 KB stages, 9 mfma per 4 KB");
        run<14, 2, 9>(src, out, nstages, grid, "4 KB stages, NO barrier (timing only)");
    }
    return 0;
}
