// do the two ways of writing the pad8 model give the same bits? (probe; build: hipcc --offload-arch=gfx950 -O3 eval_forms.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ inline float eval_a(float t, const float* x)
{
    return x[0] * expf(-t * x[1]) + x[2] + x[3] * sinf(2.0f * t) + x[4] * cosf(2.0f * t) + x[5] * sinf(5.0f * t)
         + x[6] * cosf(5.0f * t) + x[7] * t;
}
__device__ inline void basis(float t, float* b) { b[0] = sinf(2.0f * t); b[1] = cosf(2.0f * t); b[2] = sinf(5.0f * t); b[3] = cosf(5.0f * t); }
__device__ inline float eval_b(float t, const float* b, const float* x)
{
    return x[0] * expf(-t * x[1]) + x[2] + x[3] * b[0] + x[4] * b[1] + x[5] * b[2] + x[6] * b[3] + x[7] * t;
}
__global__ void k_table(const float* t, float* tab, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) basis(t[i], tab + 4 * i); }
__global__ void k_cmp(const float* t, const float* x, const float* tab, int n, int* diff)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p[8];
    for (int k = 0; k < 8; ++k) p[k] = x[8 * (i % 1024) + k];
    const float ra = eval_a(t[i], p);
    float b[4]; basis(t[i], b);
    const float rb = eval_b(t[i], b, p);
    const float rc = eval_b(t[i], tab + 4 * i, p);
    if (__float_as_uint(ra) != __float_as_uint(rb)) atomicAdd(diff, 1);
    if (__float_as_uint(ra) != __float_as_uint(rc)) atomicAdd(diff + 1, 1);
    // central difference in parameter 3, both forms
    float q[8]; for (int k = 0; k < 8; ++k) q[k] = p[k];
    q[3] = p[3] + 0.00048828125f; const float fa = eval_a(t[i], q), fb = eval_b(t[i], b, q);
    q[3] = p[3] - 0.00048828125f; const float ga = eval_a(t[i], q), gb = eval_b(t[i], b, q);
    if (__float_as_uint(fa - ga) != __float_as_uint(fb - gb)) atomicAdd(diff + 2, 1);
}
int main()
{
    const int n = 1 << 20;
    std::vector<float> t(n), x(8 * 1024);
    srand(1);
    for (auto& v : t) v = 4.0f * rand() / RAND_MAX;
    for (auto& v : x) v = 2.0f * rand() / RAND_MAX - 0.5f;
    float *dt, *dx, *tab; int* dd;
    hipMalloc(&dt, 4 * n); hipMalloc(&dx, 4 * x.size()); hipMalloc(&tab, 16 * n); hipMalloc(&dd, 16);
    hipMemcpy(dt, t.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), 4 * x.size(), hipMemcpyHostToDevice);
    hipMemset(dd, 0, 16);
    k_table<<<n / 256, 256>>>(dt, tab, n);
    k_cmp<<<n / 256, 256>>>(dt, dx, tab, n, dd);
    int h[4]; hipMemcpy(h, dd, 16, hipMemcpyDeviceToHost);
    printf("inline-vs-basis-in-registers %d, inline-vs-table %d, central difference %d of %d\n", h[0], h[1], h[2], n);
    return 0;
}
