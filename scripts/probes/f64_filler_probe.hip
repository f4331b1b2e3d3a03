// f64_filler_probe.hip -- what issues for free beside a back-to-back chain of v_mfma_f64_16x16x4_f64 on gfx950?
// One workgroup of 256 threads (one wave per SIMD). Each wave repeats { 1 MFMA ; K filler instructions } and the table is
// cycles per group for K = 0, 4, 8, 16 and several filler kinds (all fillers independent of the MFMA chain; four chains).
// Companion of f64_coexec_probe.hip (two waves per SIMD). Also: accuracy of v_rcp_f64 with 0 / 1 / 2 Newton steps.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_filler scripts/probes/f64_filler_probe.hip && /tmp/f64_filler
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef __attribute__((ext_vector_type(4))) double Acc;

template <int KIND, int K>
__global__ __launch_bounds__(256) void k(double* sink, long long* cyc, int reps)
{
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1; lds[threadIdx.x + 512] = 2; lds[threadIdx.x + 768] = 3;
    double a = 1.0 + lane * 1e-9, b = 1.0000001;
    Acc c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    double v[4] = {a, a + 1, a + 2, a + 3};
    float f[4] = {(float)a, 2.f, 3.f, 4.f};
    int q[4] = {lane, lane + 1, lane + 2, lane + 3};
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < K; ++e) {
                if constexpr (KIND == 0) v[e & 3] = fma(v[e & 3], b, a);                              // v_fma_f64
                if constexpr (KIND == 1) f[e & 3] = fmaf(f[e & 3], 1.0000001f, 0.5f);                 // v_fma_f32
                if constexpr (KIND == 2) q[e & 3] = q[e & 3] * 3 + (q[(e + 1) & 3] >> 1);              // integer VALU
                if constexpr (KIND == 3) q[e & 3] = __builtin_amdgcn_mov_dpp(q[e & 3], 0xB1, 0xF, 0xF, true) + 1;   // DPP move + add
                if constexpr (KIND == 4) v[e & 3] += lds[(q[e & 3] + 64 * e) & 1023];                  // ds_read_b64 + v_add_f64
                if constexpr (KIND == 5) v[e & 3] = __builtin_amdgcn_rcp(v[e & 3] + 1.5);              // v_rcp_f64 (+ add)
                if constexpr (KIND == 6) v[e & 3] = rint(v[e & 3] * 1.4426950408889634);               // v_mul_f64 + v_rndne_f64
                if constexpr (KIND == 7) v[e & 3] = ldexp(v[e & 3], q[e & 3] & 3);                     // v_ldexp_f64
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = c0[0] + c1[0] + v[0] + v[1] + v[2] + v[3] + f[0] + f[1] + f[2] + f[3] + q[0] + q[1] + q[2] + q[3];
}

__global__ void k_rcp(const double* d, double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    double q0 = __builtin_amdgcn_rcp(x);
    double q1 = fma(fma(-x, q0, 1.0), q0, q0);
    double q2 = fma(fma(-x, q1, 1.0), q1, q1);
    out[3 * i] = q0; out[3 * i + 1] = q1; out[3 * i + 2] = q2;
}

template <int KIND, int K> double run(double* sink, long long* cyc, int reps)
{
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, K>), dim3(1), dim3(256), 0, 0, sink, cyc, reps);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    long long w = 0;
    for (int i = 0; i < 4; ++i) w = h[i] > w ? h[i] : w;
    return (double)w / reps / 8;
}
template <int KIND> void row(const char* name, double* sink, long long* cyc, int reps)
{
    printf("%-44s K=0 %6.1f   K=4 %6.1f   K=8 %6.1f   K=16 %6.1f\n", name, run<KIND, 0>(sink, cyc, reps), run<KIND, 4>(sink, cyc, reps),
           run<KIND, 8>(sink, cyc, reps), run<KIND, 16>(sink, cyc, reps));
}

int main()
{
    double* sink; long long* cyc;
    hipMalloc(&sink, 256 * sizeof(double)); hipMalloc(&cyc, 4 * sizeof(long long));
    const int reps = 2000;
    printf("cycles per { 1 v_mfma_f64_16x16x4_f64 ; K fillers } (one wave per SIMD)\n");
    row<0>("v_fma_f64", sink, cyc, reps);
    row<1>("v_fma_f32", sink, cyc, reps);
    row<2>("integer VALU (mul + shift + add)", sink, cyc, reps);
    row<3>("v_mov_b32 dpp + v_add_u32", sink, cyc, reps);
    row<4>("ds_read_b64 + v_add_f64", sink, cyc, reps);
    row<5>("v_add_f64 + v_rcp_f64", sink, cyc, reps);
    row<6>("v_mul_f64 + v_rndne_f64", sink, cyc, reps);
    row<7>("v_ldexp_f64 (+ v_and)", sink, cyc, reps);
    // accuracy of the reciprocal over d = e + 1, e = exp(t), t in [0, 40]
    const int n = 1 << 20;
    double* hd = new double[n]; double* ho = new double[3 * n];
    for (int i = 0; i < n; ++i) hd[i] = exp(40.0 * (i + 0.5) / n) + 1.0;
    double *dd, *dout;
    hipMalloc(&dd, n * 8); hipMalloc(&dout, 3 * n * 8);
    hipMemcpy(dd, hd, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rcp, dim3(n / 256), dim3(256), 0, 0, dd, dout, n);
    hipMemcpy(ho, dout, 3 * n * 8, hipMemcpyDeviceToHost);
    double worst[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int s = 0; s < 3; ++s) {
            const long double ex = 1.0L / (long double)hd[i];
            const double rel = (double)fabsl(((long double)ho[3 * i + s] - ex) / ex);
            if (rel > worst[s]) worst[s] = rel;
        }
    printf("v_rcp_f64 max relative error: raw %.3e   1 Newton step %.3e   2 steps %.3e   (2^-53 = 1.11e-16)\n", worst[0], worst[1], worst[2]);
    return 0;
}
