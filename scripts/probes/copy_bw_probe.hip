// copy_bw_probe.hip -- what does a 50 % read / 50 % write stream reach on this part? (k_jtj_fdp reads the 1 GB difference panel and
// writes the 1 GB Jacobian: its HBM ceiling is a COPY's, not a read's.)  1 GiB -> 1 GiB, 16-byte accesses, plain and non-temporal,
// a few grid shapes; and a read-only sum for comparison.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_bw scripts/probes/copy_bw_probe.hip && /tmp/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const v2* __restrict__ src, v2* __restrict__ dst, size_t n, int per)
{
    // a workgroup owns a contiguous range, `per` elements per thread in flight
    const size_t chunk = (size_t)blockDim.x * per;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        v2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            const size_t ic = i < n ? i : n - 1;
            v[u] = NT ? __builtin_nontemporal_load(src + ic) : src[ic];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
        }
    }
}

__global__ __launch_bounds__(256) void k_read(const v2* __restrict__ src, double* out, size_t n)
{
    double s = 0;
    const size_t chunk = (size_t)blockDim.x * 8;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        v2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
            v[u] = __builtin_nontemporal_load(src + (i < n ? i : n - 1));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
    }
    if (s == 1.2345e300) out[0] = s;
}

// the access pattern of k_tanh_linear at n = 128: a wave reads 4 rows of 1 KB per step, lane (q, p) the 16 bytes p of quarter c of
// row q -- one instruction touches four 256-byte segments 1 KB apart; a contiguous range of steps per wave, RING steps in flight
template <int RING>
__global__ __launch_bounds__(256) void k_read_rows4(const v2* __restrict__ src, double* out, size_t n)
{
    const int lane = threadIdx.x & 63, q = lane >> 4, p = lane & 15;
    const size_t steps = n / 256;                                   // 4 KB = 256 v2 per step
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t per = (steps + nwaves - 1) / nwaves, s0 = wave * per < steps ? wave * per : steps, s1 = s0 + per < steps ? s0 + per : steps;
    double s = 0;
    for (size_t st = s0; st < s1; st += RING) {
        v2 v[RING][4];
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            const size_t sc = st + r < s1 ? st + r : s1 - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[r][c] = __builtin_nontemporal_load(src + sc * 256 + q * 64 + c * 16 + p);
        }
#pragma unroll
        for (int r = 0; r < RING; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) s += v[r][c].x + v[r][c].y;
    }
    if (s == 1.2345e300) out[0] = s;
}

int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(v2);
    v2 *a, *b; double* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 8);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
    };
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        const float t0 = time([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, a, b, n, 8); });
        const float t1 = time([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, a, b, n, 8); });
        const float t2 = time([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, o, n); });
        printf("grid %5d: copy 1 GiB -> 1 GiB  plain %.3f ms = %.2f TB/s   non-temporal %.3f ms = %.2f TB/s   |   read 1 GiB %.3f ms = %.2f TB/s\n",
               grid, t0, 2.0 * bytes / t0 / 1e9, t1, 2.0 * bytes / t1 / 1e9, t2, 1.0 * bytes / t2 / 1e9);
    }
    for (int grid : {256 * 4, 256 * 8, 256 * 16}) {
        const float t1 = time([&] { hipLaunchKernelGGL(k_read_rows4<1>, dim3(grid), dim3(256), 0, 0, a, o, n); });
        const float t4 = time([&] { hipLaunchKernelGGL(k_read_rows4<4>, dim3(grid), dim3(256), 0, 0, a, o, n); });
        printf("grid %5d: read 1 GiB in the 4-rows-per-step pattern of k_tanh_linear, contiguous range per wave: 1 step in flight %.3f ms = %.2f TB/s   4 steps %.3f ms = %.2f TB/s\n",
               grid, t1, 1.0 * bytes / t1 / 1e9, t4, 1.0 * bytes / t4 / 1e9);
    }
    const float tm = time([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpy device to device: %.3f ms = %.2f TB/s (read + write)\n", tm, 2.0 * bytes / tm / 1e9);
    return 0;
}
