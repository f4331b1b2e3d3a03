// f64_coexec_probe.hip -- do f64 VALU instructions and f64 MFMAs of the two waves that share a SIMD overlap on gfx950?
// One workgroup of 512 threads (waves w and w + 4 share SIMD w % 4). Each wave runs a role for `reps` rounds between two
// s_barriers and stamps s_memtime; the table is the slowest wave's cycles per round.
//   role M: 32 v_mfma_f64_16x16x4_f64 in two independent accumulator chains
//   role V: 128 v_fma_f64 in four independent chains          role T: the workload's dtanh on 4 values
//   role -: idle (waits at the barrier)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_coexec scripts/probes/f64_coexec_probe.hip && /tmp/f64_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef __attribute__((ext_vector_type(4))) double Acc;

__device__ inline double dtanh_wl(double x)     // the residual's tanh (csrc/workloads_device.h)
{
    const double ax = fabs(x);
    const double t = fmin(2.0 * ax, 40.0);
    const double kf = rint(t * 1.4426950408889634);
    double r = fma(kf, -6.93147180369123816490e-01, t);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double e = ldexp(p, (int)kf);
    const double d = e + 1.0;
    double q = __builtin_amdgcn_rcp(d);
    q = fma(fma(-d, q, 1.0), q, q);
    q = fma(fma(-d, q, 1.0), q, q);
    return copysign(fma(-2.0, q, 1.0), x);
}

// roles[w] in {0: idle, 1: MFMA, 2: FMA, 3: tanh}
__global__ __launch_bounds__(512) void k(const int* roles, double* sink, long long* cyc, int reps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int role = roles[wave];
    double a = 1.0 + lane * 1e-9, b = 1.0000001;
    Acc c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    double v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    __syncthreads();
    const long long t0 = clock64();
    if (role == 1) {
        for (int i = 0; i < reps; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
            }
        }
    } else if (role == 2) {
        for (int i = 0; i < reps; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                v0 = fma(v0, b, a); v1 = fma(v1, b, a); v2 = fma(v2, b, a); v3 = fma(v3, b, a);
            }
        }
    } else if (role == 3) {
        for (int i = 0; i < reps; ++i) {
            v0 = dtanh_wl(v0 + 0.3); v1 = dtanh_wl(v1 - 0.2); v2 = dtanh_wl(v2 + 0.1); v3 = dtanh_wl(v3 - 0.4);
        }
    }
    const long long t1 = clock64();
    __syncthreads();
    if (lane == 0) cyc[wave] = role ? t1 - t0 : 0;
    sink[threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3] + v0 + v1 + v2 + v3;
}

int main()
{
    int* roles; double* sink; long long* cyc;
    hipMalloc(&roles, 8 * sizeof(int)); hipMalloc(&sink, 512 * sizeof(double)); hipMalloc(&cyc, 8 * sizeof(long long));
    const int reps = 2000;
    struct Case { const char* name; int r[8]; int per_round_m, per_round_v; };
    const Case cases[] = {
        {"M - (MFMA on waves 0-3, partner idle)        ", {1, 1, 1, 1, 0, 0, 0, 0}},
        {"M M (both waves of a SIMD issue MFMAs)       ", {1, 1, 1, 1, 1, 1, 1, 1}},
        {"V - (f64 FMA on waves 0-3, partner idle)     ", {2, 2, 2, 2, 0, 0, 0, 0}},
        {"V V (both waves of a SIMD issue f64 FMAs)    ", {2, 2, 2, 2, 2, 2, 2, 2}},
        {"M V (one wave MFMA, its partner f64 FMA)     ", {1, 1, 1, 1, 2, 2, 2, 2}},
        {"T - (tanh x 4 on waves 0-3, partner idle)    ", {3, 3, 3, 3, 0, 0, 0, 0}},
        {"T T                                          ", {3, 3, 3, 3, 3, 3, 3, 3}},
        {"M T (one wave MFMA, its partner tanh)        ", {1, 1, 1, 1, 3, 3, 3, 3}},
        {"M on ONE SIMD only                           ", {1, 0, 0, 0, 0, 0, 0, 0}},
        {"V on ONE SIMD only                           ", {2, 0, 0, 0, 0, 0, 0, 0}},
    };
    printf("cycles per round (32 MFMA 16x16x4 f64 | 128 v_fma_f64 | 4 tanh), slowest wave of each role; reps %d\n", reps);
    for (const Case& c : cases) {
        hipMemcpy(roles, c.r, sizeof(c.r), hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, roles, sink, cyc, reps);
        hipDeviceSynchronize();
        long long h[8];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double worst[4] = {0, 0, 0, 0};
        for (int w = 0; w < 8; ++w) if (c.r[w] && (double)h[w] / reps > worst[c.r[w]]) worst[c.r[w]] = (double)h[w] / reps;
        printf("%s", c.name);
        if (worst[1] > 0) printf("  MFMA wave %8.1f (%.1f per MFMA)", worst[1], worst[1] / 32);
        if (worst[2] > 0) printf("  FMA wave %8.1f (%.2f per v_fma_f64)", worst[2], worst[2] / 128);
        if (worst[3] > 0) printf("  tanh wave %8.1f (%.1f per tanh)", worst[3], worst[3] / 4);
        printf("\n");
    }
    return 0;
}
