// unit_entries.hip -- unit-level access to the kernels of the path (parity tests, micro-benchmarks) and the standalone
// BOXCQP entry: mir_solve_box_qp_gpu_* (boxcqp.d:85-102 is D-only), mir_lsq_jtj_*, mir_lsq_fd_jtj_d, mir_lsq_fd_diff_jtj_d,
// mir_lsq_selftest_reductions. Each call owns its scratch and synchronises before it returns.
#include "driver.h"
#include "misc_kernels.h"

using namespace mirlsq;

namespace {
// Self-test of the wave reductions (mir_lsq_selftest_reductions): every wave_sum / wave_max of common.h against the plain
// butterfly on __shfl_xor (16, 32 last: the order whose pairs the DPP forms reproduce), bit for bit, on `rounds` pseudo-random
// inputs a lane. out[0..3] += mismatching lanes of sum<float>, sum<double>, max<float>, max<double>.
__global__ __launch_bounds__(256) void k_selftest_reductions(int rounds, uint32_t seed, int* out)
{
    // no contraction here: the multiplication that makes an input would be fused into the FIRST addition of whichever form
    // consumes it (one rounding less on one operand of one form), and the two forms would differ by construction
#pragma clang fp contract(off)
    uint32_t sr = seed ^ (0x9E3779B9u * (blockIdx.x * blockDim.x + threadIdx.x + 1));
    auto rnd = [&]() { sr ^= sr << 13; sr ^= sr >> 17; sr ^= sr << 5; return sr; };
    int bad[4] = {0, 0, 0, 0};
    for (int it = 0; it < rounds; ++it) {
        const float f = (float)(int32_t)rnd() * (1.0f / 65536.0f) * ((it & 7) == 0 ? 1e-20f : 1.0f);
        const double d = ((double)(int32_t)rnd() + (double)rnd() * 2.3283064365386963e-10) * ((it & 3) == 0 ? 1e-200 : 1.0);
        auto ref_sum = [](auto v) {
            v += __shfl_xor(v, 8, kWave); v += __shfl_xor(v, 4, kWave); v += __shfl_xor(v, 2, kWave); v += __shfl_xor(v, 1, kWave);
            v += __shfl_xor(v, 16, kWave); v += __shfl_xor(v, 32, kWave);
            return v;
        };
        auto ref_max = [](auto v) {
            for (int m : {8, 4, 2, 1, 16, 32}) { const auto o = __shfl_xor(v, m, kWave); v = o > v ? o : v; }
            return v;
        };
        const float sf = wave_sum(f), rf = ref_sum(f);
        const double sd = wave_sum(d), rdd = ref_sum(d);
        bad[0] += __float_as_uint(sf) != __float_as_uint(rf);
        bad[1] += __double_as_longlong(sd) != __double_as_longlong(rdd);
        bad[2] += __float_as_uint(wave_max(f)) != __float_as_uint(ref_max(f));
        bad[3] += __double_as_longlong(wave_max(d)) != __double_as_longlong(ref_max(d));
    }
    for (int k = 0; k < 4; ++k) if (bad[k]) atomicAdd(out + k, bad[k]);
}

template <typename T, typename QS>
int box_qp_entry(const QS* settings, size_t n_, const T* P, const T* q, const T* l, const T* u, T* x,
                 int unconstrainedSolution, int* iterations)
{
    if (iterations) *iterations = 0;
    if (n_ == 0) return mir_box_qp_solved;
    if (!device_available()) return mir_box_qp_numericError;
    const int n = (int)n_;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t oP = take(sizeof(T) * n * n), oq = take(sizeof(T) * n), ol = take(sizeof(T) * n), ou = take(sizeof(T) * n),
                 ox = take(sizeof(T) * n), oPm = take(sizeof(T) * n * n), oA = take(sizeof(T) * n * n),
                 oF = take(sizeof(T) * n * (n | 1)), ov = take(sizeof(T) * 12 * n), oi = take(sizeof(int32_t) * 2 * n),
                 oo = take(sizeof(int) * 4);
    char* base = nullptr;
    if (hipMalloc((void**)&base, off) != hipSuccess) return mir_box_qp_numericError;
    BoxQpArgs<T> a{};
    a.P = (T*)(base + oP); a.q = (T*)(base + oq); a.l = (T*)(base + ol); a.u = (T*)(base + ou); a.x = (T*)(base + ox);
    a.sc.Pm = (T*)(base + oPm); a.sc.A = (T*)(base + oA); a.sc.Fg = (T*)(base + oF); a.sc.vec = (T*)(base + ov);
    a.sc.ivec = (int32_t*)(base + oi); a.out = (int*)(base + oo); a.sc.dbg = nullptr;
    a.relTol = settings->relTolerance; a.absTol = settings->absTolerance; a.maxIterations = settings->maxIterations;
    a.unconstrained = unconstrainedSolution; a.n = n;
    a.f_in_lds = solve_nb(n, (int)sizeof(T)) > 0;
    int out[2] = {mir_box_qp_numericError, 0};
    bool good = hipMemcpy((void*)a.P, P, sizeof(T) * n * n, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy((void*)a.q, q, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy((void*)a.l, l, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy((void*)a.u, u, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy((void*)a.x, x, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess;
    if (good) good = launch_box_qp<T>(a, nullptr) == hipSuccess;
    if (good) {
        good = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess
            && hipMemcpy(out, a.out, sizeof(out), hipMemcpyDeviceToHost) == hipSuccess
            && hipMemcpy(x, a.x, sizeof(T) * n, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(base);
    if (!good) return mir_box_qp_numericError;
    if (iterations) *iterations = out[1];
    return out[0];
}

template <typename T>
int jtj_entry(size_t m, size_t n, T* J, const T* y, const T* y_old, const T* dx, int broyden, T* JJ, T* Jy,
              void* stream_, float* kernel_ms)
{
    if (!device_available()) return -1;
    if (n == 0 || m == 0) return -2;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const JtjPlan plan = jtj_plan<T>(m, (int)n, query_num_cu());
    const size_t packed_len = n * (n + 1) / 2 + n + 8;
    T *slabs = nullptr, *packed = nullptr, *dxdot = nullptr;
    LmState<T>* st = nullptr;
    const size_t slab_count = jtj_slab_elems(plan);
    if (hipMalloc((void**)&slabs, sizeof(T) * slab_count) != hipSuccess) return -3;
    if (hipMalloc((void**)&packed, sizeof(T) * packed_len) != hipSuccess) { (void)hipFree(slabs); return -3; }
    if (hipMalloc((void**)&st, sizeof(LmState<T>) + sizeof(T) * 8) != hipSuccess) { (void)hipFree(slabs); (void)hipFree(packed); return -3; }
    dxdot = reinterpret_cast<T*>(st + 1);
    int rc = 0;
    if (broyden) {
        // ||dx||^2 on the device (n-vector, one block)
        hipLaunchKernelGGL(k_sumsq_partial<T>, dim3(1), dim3(256), 0, stream, dx, n, packed);
        hipLaunchKernelGGL(k_sumsq_final<T>, dim3(1), dim3(256), 0, stream, packed, 1, dxdot);
    }
    JtjArgs<T> a{};
    a.J = J; a.Jout = J; a.y = y; a.y_old = y_old; a.dx = dx; a.dx_dot = dxdot; a.slabs = slabs; a.m = m; a.n = (int)n;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, stream);
    if (jtj_run<T>(plan, a, broyden != 0, packed, stream) != hipSuccess) rc = -4;
    (void)hipEventRecord(e1, stream);
    (void)jtj_unpack<T>(packed, (int)n, JJ, Jy, st, stream);
    if (hipStreamSynchronize(stream) != hipSuccess) rc = -5;
    if (kernel_ms) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); *kernel_ms = ms; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(slabs); (void)hipFree(packed); (void)hipFree(st);
    return rc;
}
int fd_jtj_entry(size_t m, size_t n, const double* Yrm, const double* twh, const double* y, double* J,
                 double* JJ, double* Jy, void* stream_, float* kernel_ms, bool diff)
{
    if (!device_available()) return -1;
    if (n == 0 || m == 0) return -2;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const JtjPlan plan = jtj_plan<double>(m, (int)n, query_num_cu());
    if (diff ? !jtj_fd_diff_ok(plan, (int)n) : (!plan.fdp && !plan.fdp8)) return -6;   // shape not covered by a fused kernel
    const size_t packed_len = n * (n + 1) / 2 + n + 8;
    double *slabs = nullptr, *packed = nullptr;
    LmState<double>* st = nullptr;
    const size_t slab_count = jtj_slab_elems(plan);
    if (hipMalloc((void**)&slabs, sizeof(double) * slab_count) != hipSuccess) return -3;
    if (hipMalloc((void**)&packed, sizeof(double) * packed_len) != hipSuccess) { (void)hipFree(slabs); return -3; }
    if (hipMalloc((void**)&st, sizeof(LmState<double>)) != hipSuccess) { (void)hipFree(slabs); (void)hipFree(packed); return -3; }
    int rc = 0;
    JtjArgs<double> a{};
    a.J = Yrm; a.Jout = J; a.y = y; a.y_old = y; a.dx = nullptr; a.dx_dot = nullptr; a.slabs = slabs; a.m = m; a.n = (int)n;
    a.twh = twh;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, stream);
    if ((diff ? jtj_run_fd_diff<double>(plan, a, packed, stream) : jtj_run_fd<double>(plan, a, packed, stream)) != hipSuccess) rc = -4;
    (void)hipEventRecord(e1, stream);
    (void)jtj_unpack<double>(packed, (int)n, JJ, Jy, st, stream);
    if (hipStreamSynchronize(stream) != hipSuccess) rc = -5;
    if (kernel_ms) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); *kernel_ms = ms; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(slabs); (void)hipFree(packed); (void)hipFree(st);
    return rc;
}
}  // namespace

extern "C" {

int mir_solve_box_qp_gpu_d(const mir_box_qp_settings_d* settings, size_t n, const double* P, const double* q,
                           const double* l, const double* u, double* x, int unconstrainedSolution, int* iterations)
{
    return box_qp_entry<double>(settings, n, P, q, l, u, x, unconstrainedSolution, iterations);
}
int mir_solve_box_qp_gpu_s(const mir_box_qp_settings_s* settings, size_t n, const float* P, const float* q,
                           const float* l, const float* u, float* x, int unconstrainedSolution, int* iterations)
{
    return box_qp_entry<float>(settings, n, P, q, l, u, x, unconstrainedSolution, iterations);
}

int mir_lsq_jtj_d(size_t m, size_t n, double* J, const double* y, const double* y_old, const double* dx, int broyden,
                  double* JJ, double* Jy, void* stream, float* kernel_ms)
{
    return jtj_entry<double>(m, n, J, y, y_old, dx, broyden, JJ, Jy, stream, kernel_ms);
}
int mir_lsq_fd_jtj_d(size_t m, size_t n, const double* Yrm, const double* twh, const double* y, double* J,
                     double* JJ, double* Jy, void* stream_, float* kernel_ms)
{
    return fd_jtj_entry(m, n, Yrm, twh, y, J, JJ, Jy, stream_, kernel_ms, false);
}
int mir_lsq_fd_diff_jtj_d(size_t m, size_t n, const double* Drm, const double* twh, const double* y, double* J,
                          double* JJ, double* Jy, void* stream_, float* kernel_ms)
{
    return fd_jtj_entry(m, n, Drm, twh, y, J, JJ, Jy, stream_, kernel_ms, true);
}
int mir_lsq_jtj_s(size_t m, size_t n, float* J, const float* y, const float* y_old, const float* dx, int broyden,
                  float* JJ, float* Jy, void* stream, float* kernel_ms)
{
    return jtj_entry<float>(m, n, J, y, y_old, dx, broyden, JJ, Jy, stream, kernel_ms);
}
int mir_lsq_selftest_reductions(int rounds, int mismatches[4])
{
    if (!mismatches || rounds <= 0) return -1;
    if (!device_available()) return -2;
    int* d = nullptr;
    if (hipMalloc((void**)&d, 4 * sizeof(int)) != hipSuccess) return -3;
    bool ok = hipMemset(d, 0, 4 * sizeof(int)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_selftest_reductions, dim3(2048), dim3(256), 0, nullptr, rounds, 12345u, d);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess
          && hipMemcpy(mismatches, d, 4 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    return ok ? 0 : -4;
}

}  // extern "C"
