// jtj_kernel.h -- fused  [Broyden rank-1 update]  +  J^T J (lower)  +  J^T y  over a tall-skinny,
// row-major J[m x n] (n <= 128), one pass over HBM.
//
// Replaces, on the reference's hot path (/root/reference/source/mir/optim/least_squares.d):
//   LS:1003-1006  axpy / gemv / scal / ger   (Broyden update, row-local)
//   LS:1052       gemv(J^T, y) -> Jy
//   LS:1065       syrk(Lower, J^T) -> JJ
//
// MI355X mapping (see DESIGN.md "jtj kernel"):
//   * a wave is an independent worker over a contiguous range of 4-row groups; no LDS and no
//     barrier in the main loop. Lane l = (q = l >> 4, p = l & 15) holds J[4g + q][16c + p] for
//     every 16-column block c -- which is exactly the A *and* B operand layout of
//     v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 for the product J_c^T J_c', so fragments go
//     HBM -> VGPR -> MFMA with no shuffle. Each 16-lane group reads one full 128-byte line.
//   * the lower triangle of 16x16 output blocks (NCB (NCB+1)/2 accumulators) lives in registers.
//     When it would not fit one wave's budget (n = 128 in f64: 288 VGPRs) it is split between two
//     "roles" (even/odd waves of a workgroup) that walk the same rows -- the second reader hits
//     L1/L2, HBM still sees every byte of J once -- so each wave keeps <= 18 blocks and two waves
//     fit per SIMD.
//   * Broyden: the row dot J[i,:].dx is a 16-lane shuffle reduction, the rank-1 update is applied
//     to the fragments in registers, written back with the same coalesced pattern (by role 0),
//     and the *updated* fragments feed the MFMAs and the J^T y accumulation.
//   * epilogue: the waves of a workgroup that share a role are summed through LDS in a fixed
//     order, one slab per workgroup goes to HBM, a second kernel sums the slabs in a fixed order
//     (deterministic; no float atomics) into the packed [JJ lower | Jy] buffer that the multi-GPU
//     all-reduce uses.
#pragma once

#include <type_traits>

#include "common.h"
#include "jtj_plan.h"

namespace mirlsq {

template <typename T, int NCB> __host__ __device__ constexpr int jtj_roles() { return jtj_roles_rt(NCB, 4 * (int)(sizeof(T) / 4)); }

// block b (linear index I (I+1)/2 + J) belongs to role floor(b * ROLES / NACC)-ish: contiguous
// ranges of ceil(NACC / ROLES) blocks
template <int NCB, int ROLES, int ROLE> __host__ __device__ constexpr bool jtj_owns(int b)
{
    constexpr int per = (jtj_nacc<NCB>() + ROLES - 1) / ROLES;
    return b >= ROLE * per && b < (ROLE + 1) * per;
}

template <typename T, int NCB, bool BROYDEN, int ROLES, int ROLE>
__device__ __forceinline__ void jtj_body(const JtjArgs<T>& a, T* red, int lane, int slot, int nslots, int wave_in_role)
{
    using Acc = typename Mma<T>::Acc;
    constexpr int NACC = jtj_nacc<NCB>();
    const int q = lane >> 4, p = lane & 15;
    const size_t m = a.m;
    const int n = a.n;

    // contiguous range of 4-row groups for this slot (all roles of a slot walk the same rows)
    const size_t G = (m + 3) / 4;
    const size_t per = (G + nslots - 1) / nslots;
    const size_t g0 = (size_t)slot * per < G ? (size_t)slot * per : G;
    const size_t g1 = g0 + per < G ? g0 + per : G;

    Acc acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = Acc{0, 0, 0, 0};
    T jy[NCB];
    int coff[NCB];
    bool cok[NCB];
    T dxr[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        jy[c] = 0;
        const int col = 16 * c + p;
        cok[c] = col < n;
        coff[c] = cok[c] ? col : n - 1;
        dxr[c] = 0;
        if constexpr (BROYDEN) { const T t = a.dx[coff[c]]; dxr[c] = cok[c] ? t : T(0); }
    }
    T neg_d = 0;
    if constexpr (BROYDEN) neg_d = -(T(1) / *a.dx_dot);

    // a fragment holds what was LOADED; rows past m and columns past n are zeroed when it is used (a select right behind a load waits
    // for the load where it is issued)
    struct Frag { T v[NCB]; T y, yo; bool rok; };

    auto load = [&](size_t g, Frag& f) {
        const size_t row = 4 * g + q;
        f.rok = row < m;
        const size_t rc = f.rok ? row : m - 1;
        const T* rp = a.J + rc * (size_t)n;
#pragma unroll
        for (int c = 0; c < NCB; ++c) f.v[c] = rp[coff[c]];
        f.y = a.y[rc];
        f.yo = 0;
        if constexpr (BROYDEN) f.yo = a.y_old[rc];
    };

    auto compute = [&](size_t g, Frag& f) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) f.v[c] = (f.rok && cok[c]) ? f.v[c] : T(0);
        f.y = f.rok ? f.y : T(0);
        if constexpr (BROYDEN) f.yo = f.rok ? f.yo : T(0);
        if constexpr (BROYDEN) {
            // LS:1003-1006 for the 4 rows of this group
            T part = 0;
#pragma unroll
            for (int c = 0; c < NCB; ++c) part += f.v[c] * dxr[c];
            part = sum16(part);
            const T t = (f.yo - f.y) + part;     // axpy(-1, y, mBuffer); gemv(1, J, dx, 1, mBuffer)
            const T u = neg_d * t;               // scal(-d, mBuffer)
            const size_t row = 4 * g + q;
            const bool rok = row < m;
            T* wp = a.Jout + (rok ? row : m - 1) * (size_t)n;
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                f.v[c] = f.v[c] + u * dxr[c];    // ger(1, mBuffer, dx, J)
                if constexpr (ROLE == 0) { if (rok && cok[c]) wp[coff[c]] = f.v[c]; }
            }
        }
        if constexpr (ROLE == 0) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) jy[c] += f.v[c] * f.y;          // LS:1052
        }
#pragma unroll
        for (int I = 0; I < NCB; ++I)
#pragma unroll
            for (int Jb = 0; Jb <= I; ++Jb)
                if (jtj_owns<NCB, ROLES, ROLE>(I * (I + 1) / 2 + Jb))
                    acc[I * (I + 1) / 2 + Jb] = Mma<T>::mma(f.v[I], f.v[Jb], acc[I * (I + 1) / 2 + Jb]);   // LS:1065
    };

    // Ring of kRing fragment sets, kAhead row groups in flight beyond the one being used; the loads of a group are issued
    // unconditionally (past the end the last group is read again): under `if` the compiler cannot count the loads in flight and
    // waits for all of them at the first use (the same change took k_jtj_wide from 0.16 to 0.56 of the MFMA peak)
    if constexpr (BROYDEN) {
        // The Broyden REWRITE variant (a reference path: MIR_LSQ_VARIANT_BROYDEN_REWRITE) keeps its original pipeline: one group ahead,
        // every load WAITED FOR where it is issued. With several roles a row group is read by every role and rewritten in place by
        // role 0: that is safe only while a role's read of group g + 1 is complete before role 0, one group further on, stores it --
        // asynchronous loads in a deeper ring let the store overtake the read (seen in test_broyden_fused_update[30000-128] and
        // test_lowrank_broyden_matches_rewriting_kernels[9973-100] the day the ring was tried here).
        auto load_now = [&](size_t g, Frag& f) {
            load(g, f);
#pragma unroll
            for (int c = 0; c < NCB; ++c) asm volatile("" : "+v"(f.v[c]));          // the values are in registers before anything else happens
        };
        // ROLES > 1: a workgroup barrier per group makes it certain -- every wave of the workgroup runs the same `per` trips (guarded
        // past its own range), role 0 stores group g + 1 in trip g + 1, which no wave enters before all have finished trip g and
        // with it their (waited-for) load of group g + 1.
        Frag fa, fb;
        if (g0 < g1) load_now(g0, fa);
        for (size_t it = 0; it < per; it += 2) {
            if constexpr (ROLES > 1) __syncthreads();
            const size_t g = g0 + it;
            if (g < g1) {
                if (g + 1 < g1) load_now(g + 1, fb);
                compute(g, fa);
            }
            if constexpr (ROLES > 1) __syncthreads();
            if (g + 1 < g1) {
                if (g + 2 < g1) load_now(g + 2, fa);
                compute(g + 1, fb);
            }
        }
    } else {
        constexpr int kAhead = 2, kRing = kAhead + 1;
        Frag fr_[kRing];
        if (g0 < g1) {
            auto issue = [&](size_t g, auto B) { load(g < g1 ? g : g1 - 1, fr_[decltype(B)::value]); };
            static_for<kAhead>([&](auto U) { issue(g0 + decltype(U)::value, U); });
            for (size_t gb = g0; gb < g1; gb += kRing) {
                static_for<kRing>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    if (gb + u < g1) {
                        issue(gb + u + kAhead, IntC<(u + kAhead) % kRing>{});
                        compute(gb + u, fr_[u]);
                    }
                });
            }
        }
    }

    if constexpr (ROLE == 0) {
        // J^T y: fold the 4 row sub-groups (lanes l, l^16, l^32, l^48 hold the same column)
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            jy[c] += wave_shfl_xor(jy[c], 16);
            jy[c] += wave_shfl_xor(jy[c], 32);
        }
    }

    // ---- workgroup reduction through LDS: the kJtjWaves / ROLES waves of this role, fixed order
    constexpr int SL = jtj_slab_len<NCB>();
    constexpr int WPR = kJtjWaves / ROLES;       // waves per role
    auto put = [&](T* dst) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            if (jtj_owns<NCB, ROLES, ROLE>(i)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(i * 4 + r) * kWave + lane] = acc[i][r];
            }
        if constexpr (ROLE == 0) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) dst[(NACC * 4 + c) * kWave + lane] = jy[c];
        }
    };
    auto add = [&](const T* src) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            if (jtj_owns<NCB, ROLES, ROLE>(i)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] += src[(i * 4 + r) * kWave + lane];
            }
        if constexpr (ROLE == 0) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) jy[c] += src[(NACC * 4 + c) * kWave + lane];
        }
    };
    // tree over wave_in_role: (0 += 2, 1 += 3), then 0 += 1  [WPR = 4]; 0 += 1 [WPR = 2]
    if constexpr (WPR == 4) {
        if (wave_in_role >= 2) put(red + (size_t)(wave_in_role - 2) * SL);
        __syncthreads();
        if (wave_in_role < 2) add(red + (size_t)wave_in_role * SL);
        __syncthreads();
        if (wave_in_role == 1) put(red);
        __syncthreads();
        if (wave_in_role == 0) add(red);
    } else if constexpr (WPR == 2) {
        // the two roles use disjoint parts of one slab image
        if (wave_in_role == 1) put(red);
        __syncthreads();
        if (wave_in_role == 0) add(red);
    }
    if (wave_in_role == 0) put(a.slabs + (size_t)blockIdx.x * SL);
}

template <typename T, int NCB> constexpr int jtj_min_waves()
{
    constexpr int regs = jtj_nacc<NCB>() * 4 * (int)(sizeof(T) / 4) / jtj_roles<T, NCB>();
    return regs > 40 ? 2 : 4;
}

template <typename T, int NCB, bool BROYDEN>
__global__ __launch_bounds__(kJtjWaves * kWave, (jtj_min_waves<T, NCB>()))
void k_jtj(JtjArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);
    constexpr int ROLES = jtj_roles<T, NCB>();
    constexpr int WPR = kJtjWaves / ROLES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave % ROLES, wir = wave / ROLES;
    const int slot = blockIdx.x * WPR + wir, nslots = gridDim.x * WPR;
    if constexpr (ROLES == 1) {
        jtj_body<T, NCB, BROYDEN, 1, 0>(a, red, lane, slot, nslots, wir);
    } else if constexpr (ROLES == 2) {
        if (role == 0) jtj_body<T, NCB, BROYDEN, 2, 0>(a, red, lane, slot, nslots, wir);
        else jtj_body<T, NCB, BROYDEN, 2, 1>(a, red, lane, slot, nslots, wir);
    } else {
        if (role == 0) jtj_body<T, NCB, BROYDEN, 4, 0>(a, red, lane, slot, nslots, wir);
        else if (role == 1) jtj_body<T, NCB, BROYDEN, 4, 1>(a, red, lane, slot, nslots, wir);
        else if (role == 2) jtj_body<T, NCB, BROYDEN, 4, 2>(a, red, lane, slot, nslots, wir);
        else jtj_body<T, NCB, BROYDEN, 4, 3>(a, red, lane, slot, nslots, wir);
    }
}

// Sum the per-workgroup slabs in a fixed order and scatter into the packed buffer
//   packed[ i (i + 1) / 2 + j ] = (J^T J)_{ij}, j <= i ;  packed[ n (n + 1) / 2 + j ] = (J^T y)_j
// blockDim = 1024 = 32 slab elements x 32 slab ranges (8 loads in flight per thread, at most two dependent round trips).
template <typename T>
__global__ __launch_bounds__(1024) void k_jtj_slab_reduce(const T* __restrict__ slabs, int nslabs, int slab_len,
                                                          int ncb, int n, T* __restrict__ packed,
                                                          T* JJ = nullptr, T* Jy = nullptr)
{
    // JJ != nullptr (single GPU: no all-reduce of `packed` follows): the expansion k_unpack_grad would do happens here -- both
    // triangles of J^T J and J^T y are written directly (the solve kernel takes max |J^T y| itself, LS:1053)
    constexpr int RANGES = 32;                               // blockDim = 1024 = 32 entries x 32 slab ranges
    __shared__ T part[RANGES][33];
    const int es = threadIdx.x & 31, sp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + es;
    T s = 0;
    if (e < slab_len) {
        const int per = (nslabs + RANGES - 1) / RANGES;
        const int b0 = sp * per, b1 = (b0 + per < nslabs) ? b0 + per : nslabs;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += slabs[(size_t)b * slab_len + e];
    }
    part[sp][es] = s;
    __syncthreads();
    if (sp == 0 && e < slab_len) {
        T tot = part[0][es];
#pragma unroll
        for (int k = 1; k < RANGES; ++k) tot += part[k][es];
        const int nacc = ncb * (ncb + 1) / 2;
        const int reg = e / kWave, lane = e % kWave;
        if (reg < nacc * 4) {
            const int blk = reg >> 2, r = reg & 3;
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= blk) ++I;
            const int Jb = blk - I * (I + 1) / 2;
            const int row = 16 * I + Mma<T>::row(lane, r);
            const int col = 16 * Jb + (lane & 15);
            if (row < n && col <= row) {
                packed[(size_t)row * (row + 1) / 2 + col] = tot;
                if (JJ) { JJ[(size_t)row * n + col] = tot; JJ[(size_t)col * n + row] = tot; }
            }
        } else {
            const int c = reg - nacc * 4;
            const int col = 16 * c + lane;
            if (lane < 16 && col < n) {
                packed[(size_t)n * (n + 1) / 2 + col] = tot;
                if (JJ) Jy[col] = tot;
            }
        }
    }
}

// address-space-qualified pointers for global_load_lds (the LDS-DMA of the eight-wave ring, jtj_ring8.h)
typedef __attribute__((address_space(3))) void* jtj_lds_ptr;
typedef const __attribute__((address_space(1))) void* jtj_gbl_ptr;

}  // namespace mirlsq
