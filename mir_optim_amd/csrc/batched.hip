// batched.hip -- BASELINE cfg 5: many small independent fits, one wavefront per problem (batched_kernel.h). The launch
// itself is the public device-header template launch_batched<Model> (include/mir_optim_amd_batched.hpp); this translation
// unit instantiates it for the three compiled-in models and adds the host-pointer entry, which completes problems whose
// step reaches a finite bound with the general solver (BOXCQP on the device, boxcqp.d:234-376).
// Nothing here is process-wide state: the A/B switch of the ladder and the profiling buffer travel in mir_lsq_batched_options.
#include "driver.h"
#include "launch_util.h"
#include "../../include/mir_optim_amd_batched.hpp"

using namespace mirlsq;

namespace {

inline int batched_model_nb(int model)
{
    using namespace mir_optim_amd;
    return model == kModelExpDecay ? ModelExpDecay::nb : model == kModelExp3Affine ? ModelExp3Affine::nb : ModelExpDecayPad8::nb;
}
inline int batched_model_n(int model)
{
    return model == kModelExpDecay ? 3 : ((model == kModelExp3Affine || model == kModelExpDecayPad8) ? 8 : 0);
}

int batched_launch(int model, const mir_least_squares_settings_s* S, size_t count, size_t m, float* x, const float* lower,
                   const float* upper, const float* t, size_t t_stride, const float* data, mir_least_squares_result_s* results,
                   const mir_lsq_batched_options* opt)
{
    using namespace mir_optim_amd;
    if (model == kModelExpDecay) return launch_batched<ModelExpDecay>(S, count, m, x, lower, upper, t, t_stride, data, results, opt);
    if (model == kModelExp3Affine) return launch_batched<ModelExp3Affine>(S, count, m, x, lower, upper, t, t_stride, data, results, opt);
    return launch_batched<ModelExpDecayPad8>(S, count, m, x, lower, upper, t, t_stride, data, results, opt);
}

struct BatchedFallbackCtx { const float* t; const float* d; hipStream_t stream; int model; };
void batched_fallback_f(void* vctx, size_t m, size_t n, const float* x, float* y)
{
    (void)n;
    using namespace mir_optim_amd;
    auto* c = static_cast<BatchedFallbackCtx*>(vctx);
    if (c->model == kModelExpDecay) launch_model_residual<ModelExpDecay>(c->t, c->d, x, y, m, c->stream);
    else if (c->model == kModelExp3Affine) launch_model_residual<ModelExp3Affine>(c->t, c->d, x, y, m, c->stream);
    else launch_model_residual<ModelExpDecayPad8>(c->t, c->d, x, y, m, c->stream);
}

// the options as this build understands them (struct_size-versioned like mir_lsq_gpu_options)
mir_lsq_batched_options batched_options(const mir_lsq_batched_options* opt)
{
    mir_lsq_batched_options o{};
    if (opt) std::memcpy(&o, opt, opt->struct_size < sizeof o ? opt->struct_size : sizeof o);
    o.struct_size = sizeof o;
    return o;
}
// A caller of the 0.1 interface passed a hipStream_t where the options pointer is now (same arity: it links). Its first word is
// not a struct size: anything below the two leading members or absurdly large is refused instead of being copied from.
bool batched_options_plausible(const mir_lsq_batched_options* opt)
{
    return !opt || (opt->struct_size >= 8 && opt->struct_size <= 1024);
}

}  // namespace

extern "C" {

int mir_lsq_batched_kernel_s(const mir_least_squares_settings_s* S, size_t count, size_t m, int model, float* x,
                             const float* lower, const float* upper, const float* t, size_t t_stride, const float* data,
                             mir_least_squares_result_s* results, const mir_lsq_batched_options* options)
{
    if (batched_model_n(model) == 0 || !batched_options_plausible(options)) return -1;
    if (count != 0 && !device_available()) return -2;
    const mir_lsq_batched_options o = batched_options(options);
    return batched_launch(model, S, count, m, x, lower, upper, t, t_stride, data, results, &o);
}

int mir_lsq_batched_posvx_s(size_t count, size_t n, const float* P, const float* rhs, float* x, int* info, void* stream)
{
    if (!P || !rhs || !x || !info || (n != 3 && n != 8)) return -1;
    if (count == 0) return 0;
    if (!device_available()) return -2;
    const unsigned blocks = (unsigned)std::min<size_t>(count, 8192);
    if (n == 8) hipLaunchKernelGGL(k_posvx_rows<8>, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), P, rhs, (int)count, x, info);
    else hipLaunchKernelGGL(k_posvx_rows<3>, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), P, rhs, (int)count, x, info);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int mir_optimize_least_squares_batched_s(const mir_least_squares_settings_s* S, size_t count, size_t m, int model,
                                         float* x, const float* lower, const float* upper,
                                         const float* t, size_t t_stride, const float* data,
                                         mir_least_squares_result_s* results, const mir_lsq_batched_options* options)
{
    if (!S || !x || !lower || !upper || !t || !data || !results || !batched_options_plausible(options)) return -1;
    const int n = batched_model_n(model);
    if (n == 0 || (t_stride != 0 && t_stride != m)) return -1;
    for (size_t i = 0; i < count; ++i) {       // defaults of LeastSquaresResult!T, LS:132-142
        results[i].status = mir_ls_numericError; results[i].iterations = results[i].fCalls = results[i].gCalls = 0;
        results[i].residual = Lim<float>::inf(); results[i].lambda = 0;
    }
    if (count == 0) return 0;
    // settings validation LS:934-943, common to all problems (codes reported per problem)
    int bad = 0;
    if (!(0 <= S->minStepQuality && S->minStepQuality < 1)) bad = mir_ls_badMinStepQuality;
    else if (!(0 <= S->goodStepQuality && S->goodStepQuality <= 1)) bad = mir_ls_badGoodStepQuality;
    else if (!(S->minStepQuality < S->goodStepQuality)) bad = mir_ls_badStepQuality;
    else if (!(1 <= S->lambdaIncrease && S->lambdaIncrease <= std::sqrt(FLT_MAX))) bad = mir_ls_badLambdaParams;
    else if (!(std::sqrt(FLT_MIN) <= S->lambdaDecrease && S->lambdaDecrease <= 1)) bad = mir_ls_badLambdaParams;
    if (!device_available()) return -2;
    const size_t lds = (size_t)(n + 2) * m * sizeof(float);
    if (m == 0 || lds > 160 * 1024 - 512) {
        std::fprintf(stderr, "[mir_optim_amd] batched entry: m = %zu does not fit one wave's LDS slice\n", m);
        return -3;
    }
    mir_lsq_batched_options o = batched_options(options);
    o.stream = nullptr;
    // the model's per-row basis table is part of this call's one allocation
    const size_t basis_b = (t_stride ? count : 1) * m * (size_t)batched_model_nb(model) * sizeof(float);
    const size_t tb = (t_stride ? count : 1) * m * sizeof(float), db = count * m * sizeof(float), xb = count * n * sizeof(float);
    char* base = nullptr;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o_ = off; off = align_up(off + bytes, 256); return o_; };
    const size_t ot = take(tb), od = take(db), ox = take(xb), ol = take(n * sizeof(float)), ou = take(n * sizeof(float)),
                 orr = take(count * sizeof(BatchedResult)), obasis = take(basis_b);
    if (hipMalloc((void**)&base, off) != hipSuccess) return -4;
    o.basis = basis_b ? (float*)(base + obasis) : nullptr;
    o.basis_bytes = basis_b;
    bool good = hipMemcpy(base + ot, t, tb, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(base + od, data, db, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(base + ox, x, xb, hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(base + ol, lower, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess
        && hipMemcpy(base + ou, upper, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    const float* dt = (const float*)(base + ot); const float* ddata = (const float*)(base + od); float* dx = (float*)(base + ox);
    const float* dlower = (const float*)(base + ol); const float* dupper = (const float*)(base + ou);
    mir_least_squares_result_s* dres = (mir_least_squares_result_s*)(base + orr);
    std::vector<BatchedResult> res(count);
    std::vector<float> x0(x, x + count * n);       // starts, for the fallback problems
    if (good && !bad) {
        good = batched_launch(model, S, count, m, dx, dlower, dupper, dt, t_stride, ddata, dres, &o) == 0;
        good = good && hipDeviceSynchronize() == hipSuccess
            && hipMemcpy(res.data(), dres, count * sizeof(BatchedResult), hipMemcpyDeviceToHost) == hipSuccess
            && hipMemcpy(x, dx, xb, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (good) {
        for (size_t i = 0; i < count; ++i) {
            if (bad) { results[i].status = bad; continue; }
            results[i].status = res[i].status; results[i].iterations = res[i].iterations; results[i].fCalls = res[i].fCalls;
            results[i].gCalls = res[i].gCalls; results[i].residual = res[i].residual; results[i].lambda = res[i].lambda;
            if (res[i].status == kBatchedNeedsGeneral) {
                // bounded step: complete this problem with the general solver (device callbacks, BOXCQP on the device)
                hipStream_t st = nullptr;
                if (hipStreamCreate(&st) != hipSuccess) { good = false; break; }
                BatchedFallbackCtx c{dt + (t_stride ? i * m : 0), ddata + i * m, st, model};
                mir_lsq_gpu_options go{};
                go.struct_size = sizeof go; go.flags = MIR_LSQ_DEVICE_CALLBACKS; go.stream = st;
                std::memcpy(x + i * n, x0.data() + i * n, n * sizeof(float));
                results[i] = mir_optimize_least_squares_gpu_s(S, m, n, x + i * n, lower, upper, &go, &c, batched_fallback_f,
                                                              nullptr, nullptr, nullptr, nullptr);
                (void)hipStreamDestroy(st);
            }
        }
    }
    (void)hipFree(base);
    return good ? 0 : -5;
}

}  // extern "C"

MIRLSQ_DEFINE_PRELOAD(batched)
