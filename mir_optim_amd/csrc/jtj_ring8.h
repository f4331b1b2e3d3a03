// jtj_ring8.h -- fused [Broyden] + J^T J + J^T y for 128 < n <= 256 (fp64, n % 16 == 0, m even): an LDS-DMA ring
// (`global_load_lds_dwordx4` stages, counted vmcnt, one raw s_barrier per stage) with EIGHT waves per workgroup.
//
// Replaces LS:1003-1006 (Broyden), LS:1052 (gemv^T), LS:1065 (syrk) like the narrower kernels. At n = 256 the lower
// triangle of J^T J is 136 MFMA blocks of 16 x 16 = 1088 accumulator VGPRs: eight waves ("roles") own 17 blocks each
// and walk the same rows, one workgroup per CU. The tiled-jobs kernel this replaces (jtj_wide.h: 10 jobs, each
// re-streaming its column panels, plus a separate Broyden pass) ran at 6.4 + 1.4 ms per pass for m = 1e6, n = 256.
//   roles 0, 1   issue every `global_load_lds_dwordx4` of a stage (counted vmcnt over DMA only), never store;
//   roles 2, 3   apply the rank-one update ONCE per row, in place in the LDS slot of stage s + 1 (and zero the rows
//                past m of a partial last stage), and write the rows back to HBM; they issue no load;
//   role 4       also accumulates J^T y;
//   all roles    run their MFMA chains on stage s, which was updated one iteration earlier.
// One raw s_barrier per stage of 16 rows; the DMA of stage s + 3 is issued right after barrier s into the slot
// stage s - 1 has just left (4 slots of up to 32 KB). (The four-wave ring kernels this scheme was first built and checked
// against at n <= 128 were retired in round 4: DESIGN_HISTORY.md.)
#pragma once

#include <type_traits>

#include "jtj_kernel.h"

namespace mirlsq {

template <int NCB> struct Jtj8Cfg {
    static constexpr int ROLES = 8;
    static constexpr int RS = 16;                            // rows per stage
    static constexpr int GPS = RS / 4;                       // 4-row groups per stage
    static constexpr int SLOT_BYTES = RS * 16 * NCB * 8;
    static constexpr int IPS = SLOT_BYTES / 1024;            // 1 KB DMA instructions per stage = 2 NCB
    static constexpr int OPS = IPS / 2;                      // per loading wave
    static constexpr int NS = 4;
    static constexpr int L = NS - 1;                         // stages issued ahead of the MFMA stage
    static constexpr int RING_BYTES = NS * SLOT_BYTES;
    static constexpr int YNS = 4;
    static constexpr int Y_OFF = RING_BYTES;
    static constexpr int YO_OFF = RING_BYTES + YNS * 1024;
    static constexpr int LDS_BYTES = RING_BYTES + 2 * YNS * 1024;
    static constexpr int JY_ROLE = 4;
    static_assert((L - 1) * OPS <= 63, "vmcnt is 6 bits");
};
constexpr int kJtj8Threads = 8 * kWave;

template <int NCB, int ROLE>
__device__ __forceinline__ void jtj8_body(const JtjArgs<double>& a, const bool broyden, unsigned char* smem, int lane,
                                          size_t s0, size_t S)
{
    using T = double;
    using Acc = typename Mma<T>::Acc;
    using C = Jtj8Cfg<NCB>;
    constexpr int NACC = jtj_nacc<NCB>();
    constexpr int n = 16 * NCB;
    constexpr bool LOADER = ROLE < 2;
    constexpr bool STORER = ROLE == 2 || ROLE == 3;
    constexpr int MYI = LOADER ? C::OPS : 0;
    constexpr int L = C::L;
    const int q = lane >> 4, p = lane & 15;
    const size_t m = a.m;

    const unsigned char* Jb = reinterpret_cast<const unsigned char*>(a.J);
    const size_t total = m * (size_t)n * sizeof(T);
    auto issue = [&](size_t s) {
        const size_t base = (s0 + s) * (size_t)C::RS * n * sizeof(T);
        unsigned char* slot = smem + (s % C::NS) * C::SLOT_BYTES;
#pragma unroll
        for (int k = 0; k < MYI; ++k) {
            const int ins = ROLE + 2 * k;
            size_t off = base + (size_t)(ins * 64 + lane) * 16;
            if (off + 16 > total) off = base;            // rows past m: any valid bytes (zeroed by the storers)
            __builtin_amdgcn_global_load_lds((jtj_gbl_ptr)(Jb + off), (jtj_lds_ptr)(slot + ins * 1024), 16, 0, 2 /* nt: J is swept once per pass */);
        }
    };
    const unsigned char* yb = reinterpret_cast<const unsigned char*>(a.y);
    const unsigned char* yob = reinterpret_cast<const unsigned char*>(a.y_old);
    const size_t ytotal = m * sizeof(T);
    const size_t nchunks = (S * C::RS + 127) / 128;
    size_t next_chunk = 0;
    auto issue_y = [&](size_t c) {
        size_t off = ((s0 * C::RS) + c * 128) * sizeof(T) + (size_t)lane * 16;
        if (off + 16 > ytotal) off = 0;
        __builtin_amdgcn_global_load_lds((jtj_gbl_ptr)(yb + off), (jtj_lds_ptr)(smem + C::Y_OFF + (c % C::YNS) * 1024), 16, 0, 0);
        if (broyden)
            __builtin_amdgcn_global_load_lds((jtj_gbl_ptr)(yob + off), (jtj_lds_ptr)(smem + C::YO_OFF + (c % C::YNS) * 1024), 16, 0, 0);
    };

    Acc acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = Acc{0, 0, 0, 0};
    T jy[NCB], dxr[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        jy[c] = 0;
        dxr[c] = 0;
        if constexpr (STORER) { if (broyden) dxr[c] = a.dx[16 * c + p]; }
    }
    T neg_d = 0;
    if constexpr (STORER) { if (broyden) neg_d = -(T(1) / *a.dx_dot); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the loads above retire before the counted waits

    T* const yring = reinterpret_cast<T*>(smem + C::Y_OFF);
    T* const yoring = reinterpret_cast<T*>(smem + C::YO_OFF);
    auto yindex = [&](size_t lr) { return (int)((lr >> 7) % C::YNS) * 128 + (int)(lr & 127); };

    // storers: LS:1003-1006 for this wave's row groups of stage st, in place in the LDS slot, plus the write-back;
    // without the Broyden update only a partial last stage needs work (rows past m -> 0)
    auto update = [&](size_t st) {
        T* slot = reinterpret_cast<T*>(smem + (st % C::NS) * C::SLOT_BYTES);
        const size_t row0 = (s0 + st) * C::RS;
        if (!broyden && row0 + C::RS <= m) return;
#pragma unroll
        for (int gi = ROLE - 2; gi < C::GPS; gi += 2) {
            T* rp = slot + (4 * gi + q) * n + p;
            const size_t row = row0 + 4 * gi + q;
            const bool rok = row < m;
            if (!broyden) {
                if (!rok) {
#pragma unroll
                    for (int c = 0; c < NCB; ++c) rp[16 * c] = T(0);
                }
                continue;
            }
            T v[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) v[c] = rp[16 * c];
            const int yi = yindex(st * C::RS + 4 * gi + q);
            const T y = yring[yi], yo = yoring[yi];
            T part = 0;
#pragma unroll
            for (int c = 0; c < NCB; ++c) part += v[c] * dxr[c];
            part = sum16(part);
            const T t = (yo - y) + part;                  // LS:1003-1004
            const T u = neg_d * t;                        // LS:1005
            T* wp = a.Jout + (rok ? row : m - 1) * (size_t)n + p;
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const T w = v[c] + u * dxr[c];            // LS:1006
                rp[16 * c] = rok ? w : T(0);
                if (rok) wp[16 * c] = w;
            }
        }
    };

    auto mfma_phase = [&](size_t s) {
        const T* slot = reinterpret_cast<const T*>(smem + (s % C::NS) * C::SLOT_BYTES);
        struct Grp { T v[NCB]; T y; };
        auto read = [&](int gi, Grp& g) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) g.v[c] = slot[(4 * gi + q) * n + 16 * c + p];
            g.y = 0;
            // rows past m hold zeros in the slot, so whatever finite value the y ring has there does not count
            if constexpr (ROLE == C::JY_ROLE) g.y = yring[yindex(s * C::RS + 4 * gi + q)];
        };
        auto work = [&](const Grp& g) {
            if constexpr (ROLE == C::JY_ROLE) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) jy[c] += g.v[c] * g.y;     // LS:1052
            }
#pragma unroll
            for (int I = 0; I < NCB; ++I)
#pragma unroll
                for (int Jb2 = 0; Jb2 <= I; ++Jb2)
                    if (jtj_owns<NCB, C::ROLES, ROLE>(I * (I + 1) / 2 + Jb2))
                        acc[I * (I + 1) / 2 + Jb2] = Mma<T>::mma(g.v[I], g.v[Jb2], acc[I * (I + 1) / 2 + Jb2]);   // LS:1065
        };
        Grp ga, gb;
        read(0, ga);
#pragma unroll
        for (int gi = 0; gi < C::GPS; gi += 2) {
            if (gi + 1 < C::GPS) read(gi + 1, gb);
            work(ga);
            if (gi + 1 < C::GPS) {
                if (gi + 2 < C::GPS) read(gi + 2, ga);
                work(gb);
            }
        }
    };

    // ---- prologue: L stages in flight, stage 0 landed and updated
    if constexpr (ROLE == 1) {
        while (next_chunk < 3 && next_chunk < nchunks) issue_y(next_chunk++);
    }
    const size_t pre = S < (size_t)L ? S : (size_t)L;
    if constexpr (LOADER) {
        for (size_t s = 0; s < pre; ++s) issue(s);
        if (pre == (size_t)L) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((L - 1) * C::OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                            // stage 0 (and the first y chunks) are in LDS
    if constexpr (STORER) { if (S > 0) update(0); }

    for (size_t s = 0; s < S; ++s) {
        if constexpr (ROLE == 1) {
            while (next_chunk <= ((s + 1) * C::RS) / 128 + 2 && next_chunk < nchunks) issue_y(next_chunk++);
        }
        if constexpr (LOADER) {
            // outstanding here: stages s + 1 .. s + L - 1; stage s + 1 has landed once at most L - 2 stages remain
            if (s + L - 1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((L - 2) * C::OPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (STORER) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS writes of update(s) are done
        }
        __builtin_amdgcn_s_barrier();                        // stage s is updated, stage s + 1 has landed, slot of s - 1 is free
        if constexpr (LOADER) { if (s + L < S) issue(s + L); }
        if constexpr (STORER) { if (s + 1 < S) update(s + 1); }
        mfma_phase(s);
    }

    T* dst = a.slabs + (size_t)blockIdx.x * jtj_slab_len<NCB>();
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        if (jtj_owns<NCB, C::ROLES, ROLE>(i)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(i * 4 + r) * kWave + lane] = acc[i][r];
        }
    if constexpr (ROLE == C::JY_ROLE) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            jy[c] += wave_shfl_xor(jy[c], 16);
            jy[c] += wave_shfl_xor(jy[c], 32);
            dst[(NACC * 4 + c) * kWave + lane] = jy[c];
        }
    }
}

template <int NCB>
__global__ __launch_bounds__(kJtj8Threads, 1) void k_jtj8(JtjArgs<double> a, int broyden)
{
    using C = Jtj8Cfg<NCB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t Stot = (a.m + C::RS - 1) / C::RS;
    const size_t per = (Stot + gridDim.x - 1) / gridDim.x;
    const size_t s0 = (size_t)blockIdx.x * per < Stot ? (size_t)blockIdx.x * per : Stot;
    const size_t s1 = s0 + per < Stot ? s0 + per : Stot;
    const size_t S = s1 - s0;
    const bool br = broyden != 0;
    switch (wave) {
    case 0: jtj8_body<NCB, 0>(a, br, smem8, lane, s0, S); break;
    case 1: jtj8_body<NCB, 1>(a, br, smem8, lane, s0, S); break;
    case 2: jtj8_body<NCB, 2>(a, br, smem8, lane, s0, S); break;
    case 3: jtj8_body<NCB, 3>(a, br, smem8, lane, s0, S); break;
    case 4: jtj8_body<NCB, 4>(a, br, smem8, lane, s0, S); break;
    case 5: jtj8_body<NCB, 5>(a, br, smem8, lane, s0, S); break;
    case 6: jtj8_body<NCB, 6>(a, br, smem8, lane, s0, S); break;
    default: jtj8_body<NCB, 7>(a, br, smem8, lane, s0, S); break;
    }
}

}  // namespace mirlsq
