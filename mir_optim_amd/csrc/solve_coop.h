// solve_coop.h -- helper workgroups for the any-n solve (solve_big.h): the n^3 / 3 flops of ?potrf and the n^2 sweeps of the
// residual / prediction products leave ONE compute unit at n > 256 with milliseconds of work per pass (profiles/r05: 6.5 ms per
// solve at n = 1024, 49 % of a whole fit). The ladder entry's workgroup ("main", peer 0) keeps every serial piece of
// boxcqp.d:122-379 / ?posvx exactly as it is and hands the wide pieces to W - 1 more workgroups of the same launch ("helpers")
// as JOBS through a few words in global memory:
//
//   main:    [fence]  desc <- job arguments;  job <- (epoch, seq)      ... its own share ...   wait done[1 .. W-1] >= (epoch, seq)  [fence]
//   helper:  wait job >= (epoch, seq + 1);  read desc  [fence]          ... its share ...       [fence]  done[p] <- (epoch, seq)
//
// Every word is written by ONE workgroup and only ever grows: (epoch << 32 | seq), epoch = the launch's number (the host counts
// solve launches per workspace), so nothing is reset between launches and a stale word of an earlier launch never satisfies a
// wait. Words and job data that cross workgroups inside a job go through agent-scope accesses (sc1: write-through stores,
// L2-bypassing loads -- the L2 of another XCD is not coherent with ours inside a kernel); jobs whose inputs were written with
// ordinary stores ask for agent-scope fences on both sides instead ([fence] above).
//
// Progress: the launch is an ordinary one (W <= 16 workgroups per ladder entry, at most 8 entries): a helper that is not resident
// yet is started as soon as any compute unit has room, and no kernel of this library waits on another one's workgroups, so the
// waits below are finite; they are bounded all the same (kCoopSpinSeconds) and a timeout surfaces as numericError.
#pragma once

#include "common.h"
#include "solve_kernel.h"

namespace mirlsq {

// (kCoopMaxPeers, kCoopLine, kCoopWords: solve_types.h -- job | done[1 .. 15] (line p) | desc (8 words) | abort)
constexpr int kCoopJobWord = 0;
constexpr int kCoopDescWord = kCoopLine * kCoopMaxPeers;
constexpr int kCoopAbortWord = kCoopLine * (kCoopMaxPeers + 1);
constexpr double kCoopSpinSeconds = 5.0;

// (kCoopPanels, kCoopBlockCols: solve_types.h -- the host sizes the scratch with them)
enum : uint32_t { kCoopExit = 1, kCoopPotrfT = 2, kCoopSymv = 3, kCoopCopy2 = 4 };
enum : uint32_t { kCoopFence = 0x100,                           // job flag: agent-scope fences around the job (see above)
                  kCoopRelease = 0x200 };                       // job flag: main writes its L2 back before publishing (its ordinary
                                                                // stores become visible to the helpers' agent-scope LOADS); nothing else

__device__ __forceinline__ void coop_st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long coop_ld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coop_st(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coop_ld(const double* p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void coop_st(float* p, float v)
{
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), (unsigned int)__float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float coop_ld(const float* p)
{
    return __int_as_float((int)__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// the same with the global address space spelled out (a generic pointer makes these FLAT instructions, which also count as LDS
// operations -- solve_kernel.h, as_global)
__device__ __forceinline__ double coop_ld(gbl_cptr<double> p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const __attribute__((address_space(1))) unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float coop_ld(gbl_cptr<float> p)
{
    return __int_as_float((int)__hip_atomic_load((const __attribute__((address_space(1))) unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void coop_st(gbl_ptr<double> p, double v)
{
    __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coop_st(gbl_ptr<float> p, float v)
{
    __hip_atomic_store((__attribute__((address_space(1))) unsigned int*)p, (unsigned int)__float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coop_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one lane's bounded wait for *p >= target
__device__ __forceinline__ bool coop_wait_ge(const unsigned long long* p, unsigned long long target, const unsigned long long* abort_word)
{
    long long t0 = 0;
    for (uint32_t spins = 0;; ++spins) {
        if (coop_ld64(p) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u) {
            if (coop_ld64(abort_word) >= (target & 0xffffffff00000000ull)) return false;      // somebody of THIS launch gave up
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if ((double)(now - t0) > kCoopSpinSeconds * 1e8) return false;
        }
    }
}

struct CoopCtx {
    unsigned long long* w;      // kCoopWords sync words of this ladder entry (zero when the workspace was created)
    int W, p;                   // peers of the entry, this workgroup's rank (0 = main)
    uint32_t epoch, seq;
    int failed;                 // a wait timed out: the caller reports a numeric error
    int* s_flag;                // one int of LDS for broadcasts
    unsigned long long* s_desc; // 8 words of LDS: the job being executed
};

// main: publish a job. d[0 .. 5]: arguments (uniform over the workgroup). Collective over the workgroup.
__device__ inline void coop_publish(CoopCtx& c, uint32_t type, const unsigned long long (&d)[6])
{
    const int tid = threadIdx.x;
    if (type & kCoopFence) __threadfence();
    else if (type & kCoopRelease) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    coop_drain();
    __syncthreads();
    ++c.seq;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) coop_st64(c.w + kCoopDescWord + k, d[k]);
        coop_st64(c.w + kCoopDescWord + 6, (unsigned long long)type);
        coop_drain();
        coop_st64(c.w + kCoopJobWord, ((unsigned long long)c.epoch << 32) | c.seq);
    }
}
// main: wait until every helper has finished the job just published. Collective.
__device__ inline void coop_collect(CoopCtx& c, uint32_t type)
{
    const int tid = threadIdx.x;
    coop_drain();
    if (tid < kWave) {
        bool good = true;
        if (tid >= 1 && tid < c.W)
            good = coop_wait_ge(c.w + kCoopLine * tid, ((unsigned long long)c.epoch << 32) | c.seq, c.w + kCoopAbortWord);
        good = __all(good);
        if (tid == 0) *c.s_flag = good ? 1 : 0;
    }
    __syncthreads();
    if (!*c.s_flag) {
        c.failed = 1;
        if (tid == 0) coop_st64(c.w + kCoopAbortWord, ((unsigned long long)c.epoch << 32) | 1u);
    }
    if (type & kCoopFence) __threadfence();
    __syncthreads();
}
// helper: wait for the next job; its arguments land in c.s_desc. Returns the type (kCoopExit after a timeout). Collective.
__device__ inline uint32_t coop_next(CoopCtx& c)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        const bool good = coop_wait_ge(c.w + kCoopJobWord, ((unsigned long long)c.epoch << 32) | (c.seq + 1), c.w + kCoopAbortWord);
        *c.s_flag = good ? 1 : 0;
    }
    __syncthreads();
    const bool good = *c.s_flag != 0;
    __syncthreads();
    if (!good) { c.failed = 1; return kCoopExit; }
    ++c.seq;
    if (tid < 7) c.s_desc[tid] = coop_ld64(c.w + kCoopDescWord + tid);
    __syncthreads();
    const uint32_t type = (uint32_t)c.s_desc[6];
    if (type & kCoopFence) __threadfence();
    return type;
}
// helper: the job is done. Collective.
__device__ inline void coop_done(CoopCtx& c, uint32_t type)
{
    if (type & kCoopFence) __threadfence();
    coop_drain();
    __syncthreads();
    if (threadIdx.x == 0) coop_st64(c.w + kCoopLine * c.p, ((unsigned long long)c.epoch << 32) | c.seq);
}

// ---- job kCoopPotrfT: ?potrf's update of a look-ahead block (kCoopBlockCols = 32 columns) AHEAD of the factorisation. potrf_big (left-looking, 16-column panels)
// spends 70 % of its time at n = 1024 on S = L[rows, :c0] L[panel, :c0]^T -- one CU's matrix cores. A job per panel would be a
// dozen memory-level round trips per panel (measured: 52 us a panel against 57 on one workgroup); instead the helpers work one
// BLOCK of kCoopPanels panels (w = 16 kCoopPanels columns) ahead: while main factors the panels of block B - 1 they form, for
// block B (columns w B .. w B + w - 1),
//     T[i][c] = sum over the columns q < w (B - 1) of L[i][q] L[w B + c][q],      rows i >= w B
// -- the part of S that only needs blocks <= B - 2, final when the job is published -- and main adds the columns of blocks
// B - 1 and B itself (at most 2 kCoopPanels - 1 panels) on its own matrix cores. One job per block, its time hidden behind
// main's panels. (Blocks of four panels: 16 jobs at n = 1024 but up to seven local panels per panel step -- 1.09 ms of the
// factorisation; of two: measured below.) Operands are agent-scope loads of F (main writes its L2 back when it publishes);
// T is stored with agent-scope stores, n x w row-major.
// d[0] = F, d[1] = T, d[2] = n | ldf << 32, d[3] = B.
template <typename T>
__device__ inline void coop_potrf_t(const unsigned long long* d, int p, int W)
{
    using Acc = typename Mma<T>::Acc;
    const gbl_cptr<T> F = as_global(reinterpret_cast<const T*>(d[0]));
    const gbl_ptr<T> Tb = as_global_w(reinterpret_cast<T*>(d[1]));
    const int n = (int)(d[2] & 0xffffffffull), ldf = (int)(d[2] >> 32), B = (int)d[3];
    const int nb = (n + 15) / 16;
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int slots = (W - 1) * waves, slot = (p - 1) * waves + wave;     // main (p = 0) takes no share: it is factoring
    const int depth = kCoopBlockCols * (B - 1);                           // columns q < depth
    if (p < 1 || depth <= 0) return;
    // a unit = a 16-row block of T with the block's two 16-column panels (two accumulators): a 16-column step of depth is 12
    // loads -- two groups of four steps fit the registers, so eight steps' loads are in flight (the operands come from memory,
    // ~2.5 us away: with one step ahead the job took longer than the panels main factors meanwhile)
    const int units = nb - kCoopPanels * B;
    const int nj = depth / 16;
    for (int un = slot; un < units; un += slots) {
        const int rb = kCoopPanels * B + un;
        const int ra_ = 16 * rb + lr, rowa = ra_ < n ? ra_ : n - 1;
        int rowb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int r = kCoopBlockCols * B + 16 * u + lr; rowb[u] = r < n ? r : n - 1; }
        Acc acc[2] = {Acc{0, 0, 0, 0}, Acc{0, 0, 0, 0}};
        T fa[2][4][4], fb[2][4][4][2];                                     // [group buffer][step in group][k-step][...]
        auto load = [&](int j0, auto BUF) {                                 // steps j0 .. j0 + 3 (clamped: always 48 loads)
            constexpr int buf = decltype(BUF)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int j = j0 + g < nj ? j0 + g : nj - 1;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const size_t col = (size_t)(16 * j + 4 * s4 + lk) * ldf;
                    fa[buf][g][s4] = coop_ld(F + rowa + col);
#pragma unroll
                    for (int u = 0; u < 2; ++u) fb[buf][g][s4][u] = coop_ld(F + rowb[u] + col);
                }
            }
        };
        auto mul = [&](int j0, auto BUF) {
            constexpr int buf = decltype(BUF)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (j0 + g < nj) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = Mma<T>::mma(fa[buf][g][s4], fb[buf][g][s4][u], acc[u]);
                }
        };
        load(0, IntC<0>{});
        for (int j0 = 0; j0 < nj; j0 += 8) {
            if (j0 + 4 < nj) load(j0 + 4, IntC<1>{});
            mul(j0, IntC<0>{});
            if (j0 + 4 < nj) {
                if (j0 + 8 < nj) load(j0 + 8, IntC<0>{});
                mul(j0 + 4, IntC<1>{});
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + Mma<T>::row(lane, r);
                if (row < n) coop_st(Tb + (size_t)row * kCoopBlockCols + 16 * u + lr, acc[u][r]);
            }
    }
}

// ---- job kCoopCopy2: dst0[i] = dst1[i] = src[i], i < count, the range dealt over all peers (main included). Ordinary loads and
// stores: published with kCoopFence. d[0] = src, d[1] = dst0, d[2] = dst1, d[3] = count.
template <typename T>
__device__ inline void coop_copy2(const unsigned long long* d, int p, int W)
{
    const gbl_cptr<T> src = as_global(reinterpret_cast<const T*>(d[0]));
    const gbl_ptr<T> d0 = as_global_w(reinterpret_cast<T*>(d[1])), d1 = as_global_w(reinterpret_cast<T*>(d[2]));
    const size_t count = (size_t)d[3];
    const size_t per = ((count + W - 1) / W + 15) / 16 * 16;
    const size_t lo = (size_t)p * per, hi = lo + per < count ? lo + per : count;
    for (size_t base = lo + threadIdx.x; base < hi; base += (size_t)16 * blockDim.x) {
        T v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const size_t idx = base + (size_t)u * blockDim.x; v[u] = src[idx < count ? idx : count - 1]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const size_t idx = base + (size_t)u * blockDim.x;
            if (idx < hi) { d0[idx] = v[u]; d1[idx] = v[u]; }
        }
    }
}

// ---- job kCoopSymv: y = M v (and optionally z = |M| |v|) for a symmetric n x n M stored full (entry (i, k) read as M[i + k ld]:
// consecutive rows in consecutive lanes), the columns dealt over the peers in contiguous ranges and each range over a
// workgroup's waves... one ROW per thread within a range: partial sums part[(peer * 2 + which) * n + i]. The caller adds the W
// partials of a row in peer order. d[0] = M, d[1] = v, d[2] = part, d[3] = n | ld << 32, d[4] = want_abs.
// (Inputs written with ordinary stores: published with kCoopFence.)
template <typename T>
__device__ inline void coop_symv(const unsigned long long* d, int p, int W, T* xs /* LDS >= columns of a range */)
{
    const T* M = reinterpret_cast<const T*>(d[0]);
    const T* v = reinterpret_cast<const T*>(d[1]);
    T* part = reinterpret_cast<T*>(d[2]);
    const int n = (int)(d[3] & 0xffffffffull), ld = (int)(d[3] >> 32);
    const bool want_abs = d[4] != 0;
    const int per = ((n + W - 1) / W + 15) / 16 * 16;          // columns of a peer's range (a multiple of 16)
    const int k_lo = p * per, k_hi = (k_lo + per < n) ? k_lo + per : n;
    __syncthreads();
    for (int k = threadIdx.x; k_lo + k < k_hi; k += blockDim.x) xs[k] = v[k_lo + k];
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x, ic = i < n ? i : n - 1;
        T ra = 0, wa = 0;
        for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
            T av[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) av[u] = M[ic + (size_t)(k0 + u < n ? k0 + u : n - 1) * ld];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + u < k_hi) { const T xv = xs[k0 + u - k_lo]; ra += av[u] * xv; if (want_abs) wa += dabs(av[u]) * dabs(xv); }
        }
        if (i < n) {
            part[(size_t)(2 * p) * n + i] = ra;
            if (want_abs) part[(size_t)(2 * p + 1) * n + i] = wa;
        }
    }
}

// helper workgroups' life: jobs until kCoopExit. Out of line and with a context of its own (nothing here escapes into the
// main workgroup's call tree). xs: LDS scratch of >= ceil(n / W) + 16 elements.
// (Out of line ON PURPOSE: inlined into k_lm_solve_big -- the context then shared a stack slot with main's, every access to it a
// scratch / FLAT one -- the build of hipcc 7.2 deadlocked: progress markers in global memory showed waves 1-7 of a helper going
// round this loop freely, through its workgroup barriers, while wave 0 polled for the next job and never stored its done word.
// As a function of its own the same source behaves.)
template <typename T>
__device__ __noinline__ void coop_helper_loop(unsigned long long* w, int W, int p, uint32_t epoch, T* xs)
{
    __shared__ int s_flag;
    __shared__ unsigned long long s_desc[8];
    CoopCtx c{w, W, p, epoch, 0u, 0, &s_flag, s_desc};
    for (;;) {
        const uint32_t type = coop_next(c);
        const uint32_t what = type & 0xffu;
        if (what == kCoopPotrfT) coop_potrf_t<T>(s_desc, p, W);
        else if (what == kCoopSymv) coop_symv<T>(s_desc, p, W, xs);
        else if (what == kCoopCopy2) coop_copy2<T>(s_desc, p, W);
        if (what == kCoopExit) return;                       // (also after a timeout: nobody waits for this workgroup's word then)
        coop_done(c, type);
    }
}

}  // namespace mirlsq
