// latency_probe.hip -- dependent-chain latencies on one wave / one 256-thread workgroup of gfx950 (cycles by s_memtime
// via clock64): f64 FMA, v_rsq_f64 + Newton (rsqrt_sqrt of common.h), DPP row broadcast of a double, LDS read -> use,
// workgroup barrier. Feeds the design of the one-workgroup solve kernel (csrc/solve_lds.h). Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o latency_probe scripts/probes/latency_probe.hip && ./latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mir_optim_amd/csrc/common.h"
using namespace mirlsq;

__global__ void k(double* out, long long* cyc, int reps)
{
    __shared__ double lds[512];
    const int tid = threadIdx.x;
    lds[tid] = 1.0 + tid * 1e-9; lds[256 + tid] = 0.5;
    __syncthreads();
    double a = out[tid], b = 1.0000001, c = 1e-9;
    long long t0, t1;
    // 1. dependent f64 FMA chain
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a = fma(a, b, c);
    }
    t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
    // 2. rsqrt_sqrt chain
    double x = a + 2.0;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { double ri, d; rsqrt_sqrt(x, ri, d); x = x + ri * 1e-3 + d * 1e-9; }
    }
    t1 = clock64();
    if (tid == 0) cyc[1] = t1 - t0;
    // 3. DPP row broadcast chain (double) + dependent add
    double y = x;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) y = dpp_row_bcast<3>(y) + c;
    }
    t1 = clock64();
    if (tid == 0) cyc[2] = t1 - t0;
    // 4. LDS read -> dependent address chain
    int idx = tid & 255;
    double acc = 0;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { const double v = lds[idx]; acc += v; idx = (idx + (v > 0.9 ? 1 : 2)) & 255; }
    }
    t1 = clock64();
    if (tid == 0) cyc[3] = t1 - t0;
    // 5. barriers
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) __syncthreads();
    }
    t1 = clock64();
    if (tid == 0) cyc[4] = t1 - t0;
    // 6. LDS write -> barrier -> read by another thread -> dependent FMA (one exchange step)
    double e = acc;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { lds[tid] = e; __syncthreads(); e = fma(lds[(tid + 17) & 255], 0.999, 1e-3); __syncthreads(); }
    }
    t1 = clock64();
    if (tid == 0) cyc[5] = t1 - t0;
    // 7. independent f64 FMAs (issue rate): 8 chains
    double f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) f[u] = e + u;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int u = 0; u < 8; ++u) f[u] = fma(f[u], b, c);
    }
    t1 = clock64();
    if (tid == 0) cyc[6] = t1 - t0;
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += f[u];
    out[tid] = a + x + y + acc + e + s;
}

int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 8 * 8);
    hipMemset(out, 0, 256 * 8);
    const int reps = 1000;
    for (int threads : {64, 256}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, cyc, reps);
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, cyc, reps);
        long long h[8];
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        // clock64 = s_memtime, 100 MHz on gfx9xx: 1 tick = 10 ns = ~24 shader cycles at 2.4 GHz
        printf("threads %d (ticks of 10 ns per op): fma64 dep %.3f  rsqrt_sqrt %.3f  dpp_bcast+add %.3f  lds dep %.3f  barrier %.3f  lds exchange step %.3f  fma64 indep %.3f\n",
               threads, h[0] / (16.0 * reps), h[1] / (4.0 * reps), h[2] / (16.0 * reps), h[3] / (16.0 * reps), h[4] / (16.0 * reps),
               h[5] / (8.0 * reps), h[6] / (16.0 * reps));
    }
    return 0;
}
