// jtj_pc32.h -- J^T J + J^T y of a given J in FLOAT, producer / consumer waves (gfx950).
//
// The f32 instantiation of the solver (mir_optimize_least_squares_s, least_squares.d:729-748; the products are LS:1052, 1065)
// used the register-streaming kernel k_jtj<float, .> for its J^T J: one 4-row group per wave in flight, 0.11 of the HBM
// peak at m = 1e6, n = 128 (profiles/r03/shape_sweep.jsonl). This is the scheme of k_jtj_fdp<NCB, false> (jtj_fdp.h) on
// v_mfma_f32_16x16x4_f32:
//   * 8 waves per workgroup, 2 workgroups per CU: 4 PRODUCERS copy 64-row stages of J into one of two LDS slots with
//     coalesced 16-byte loads (four floats; NCB loads per lane and stage in registers: 16 KB per wave in flight) and issue
//     the next stage's loads before they wait at the stage barrier; 4 CONSUMERS (the MFMA roles: a quarter of the lower
//     block triangle each) read ds_read_b32 fragments -- lane (q, p) holds J[4 g + q][16 c + p], both the A and the B
//     operand of the instruction -- run the MFMA chains, role 0 also J^T y;
//   * LDS rows are padded by 16 floats (row stride 16 NCB + 16): the four rows q of a fragment read then fall on four
//     disjoint sets of 16 banks (unpadded, a 512-byte row stride puts them on the same 16);
//   * slabs in the layout of every other J^T J kernel, so k_jtj_slab_reduce finishes the job (Mma<float>::row is the
//     f32 accumulator map). n % 4 == 0 (rows start on 16-byte boundaries), n <= 128; any m (rows past m are zeros in LDS).
// Bytes per launch 4 (m n + m), flops m n (n + 1) + 2 m n: at n = 128 both rooflines sit near 0.1 ms (8 TB/s, 157 TF).
#pragma once

#include "jtj_kernel.h"

namespace mirlsq {

template <int NCB> struct JtjPc32Cfg {
    static constexpr int N = 16 * NCB;                     // padded column count
    static constexpr int LDJ = N + 16;                     // LDS row stride (floats)
    static constexpr int RP = 16;                          // rows per producer wave and stage
    static constexpr int RS = 4 * RP;                      // rows per stage
    static constexpr int GPS = RS / 4;                     // 4-row groups per stage
    static constexpr int NI = RP * N / (4 * kWave);        // 16-byte loads per lane and stage = NCB
    static constexpr int SLOT_FLOATS = RS * LDJ + RS;      // the J stage, then its y values
    static constexpr int LDS_BYTES = 2 * SLOT_FLOATS * 4;
    static constexpr int THREADS = 8 * kWave;
};

typedef float pc32_v4 __attribute__((ext_vector_type(4)));

template <int NCB>
__device__ __forceinline__ void pc32_producer(const JtjArgs<float>& a, float* smem, int lane, int w, size_t s0, size_t S)
{
    using C = JtjPc32Cfg<NCB>;
    constexpr int NI = C::NI;
    constexpr int QP = C::N / 4;                           // quads per padded row
    const size_t m = a.m;
    const int nr = a.n;
    const size_t qr = (size_t)(nr / 4);                    // quads per source row
    const pc32_v4* __restrict__ Jp = reinterpret_cast<const pc32_v4*>(a.J);

    int prow[NI], pcol[NI];
    bool pad[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = kWave * i + lane;
        prow[i] = f / QP;
        const int jq = f % QP;
        pad[i] = 4 * jq >= nr;
        pcol[i] = pad[i] ? 0 : jq;
    }
    pc32_v4 b[NI];
    float yb = 0;
    auto issue = [&](size_t s) {
        const size_t row0 = (s0 + s) * C::RS + C::RP * (size_t)w;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            size_t row = row0 + prow[i];
            row = row < m ? row : m - 1;
            b[i] = Jp[row * qr + pcol[i]];
        }
        size_t yr = row0 + (lane & (C::RP - 1));
        yr = yr < m ? yr : m - 1;
        yb = a.y[yr];
    };
    auto convert = [&](size_t s) {
        float* slot = smem + (s & 1) * C::SLOT_FLOATS;
        const size_t row0 = (s0 + s) * C::RS + C::RP * (size_t)w;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            pc32_v4 v = b[i];
            if (pad[i] || row0 + prow[i] >= m) v = pc32_v4{0.f, 0.f, 0.f, 0.f};
            const int f = kWave * i + lane;
            *reinterpret_cast<pc32_v4*>(slot + (w * C::RP + prow[i]) * C::LDJ + 4 * (f % QP)) = v;
        }
        if (lane < C::RP) slot[C::RS * C::LDJ + C::RP * w + lane] = (row0 + lane < m) ? yb : 0.f;
    };
    if (S > 0) issue(0);
    for (size_t t = 0; t < S; ++t) {
        convert(t);
        if (t + 1 < S) issue(t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

template <int NCB, int ROLE>
__device__ __forceinline__ void pc32_consumer(const JtjArgs<float>& a, const float* smem, int lane, size_t s0, size_t S)
{
    using T = float;
    using Acc = typename Mma<T>::Acc;
    using C = JtjPc32Cfg<NCB>;
    constexpr int NACC = jtj_nacc<NCB>();
    const int q = lane >> 4, p = lane & 15;

    Acc acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = Acc{0, 0, 0, 0};
    T jy[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) jy[c] = 0;

    for (size_t s = 0; s < S; ++s) {
        __builtin_amdgcn_s_barrier();
        const T* slot = smem + (s & 1) * C::SLOT_FLOATS;
        struct Grp { T v[NCB]; T y; };
        auto read = [&](int gi, Grp& g) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) g.v[c] = slot[(4 * gi + q) * C::LDJ + 16 * c + p];
            g.y = 0;
            if constexpr (ROLE == 0) g.y = slot[C::RS * C::LDJ + 4 * gi + q];
        };
        auto work = [&](const Grp& g) {
            if constexpr (ROLE == 0) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) jy[c] += g.v[c] * g.y;     // LS:1052
            }
#pragma unroll
            for (int I = 0; I < NCB; ++I)
#pragma unroll
                for (int Jb = 0; Jb <= I; ++Jb)
                    if (jtj_owns<NCB, 4, ROLE>(I * (I + 1) / 2 + Jb))
                        acc[I * (I + 1) / 2 + Jb] = Mma<T>::mma(g.v[I], g.v[Jb], acc[I * (I + 1) / 2 + Jb]);   // LS:1065
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
    }

    T* dst = a.slabs + (size_t)blockIdx.x * jtj_slab_len<NCB>();
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        if (jtj_owns<NCB, 4, ROLE>(i)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(i * 4 + r) * kWave + lane] = acc[i][r];
        }
    if constexpr (ROLE == 0) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            jy[c] += wave_shfl_xor(jy[c], 16);
            jy[c] += wave_shfl_xor(jy[c], 32);
            dst[(NACC * 4 + c) * kWave + lane] = jy[c];
        }
    }
}

template <int NCB>
__global__ __launch_bounds__(8 * kWave) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_jtj_pc32(JtjArgs<float> a)
{
    using C = JtjPc32Cfg<NCB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char pc32_smem[];
    float* smem = reinterpret_cast<float*>(pc32_smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const size_t Stot = (a.m + C::RS - 1) / C::RS;
    const size_t per = (Stot + gridDim.x - 1) / gridDim.x;
    const size_t s0 = (size_t)blockIdx.x * per < Stot ? (size_t)blockIdx.x * per : Stot;
    const size_t s1 = s0 + per < Stot ? s0 + per : Stot;
    const size_t S = s1 - s0;

    if (wave == 0) pc32_consumer<NCB, 0>(a, smem, lane, s0, S);
    else if (wave == 1) pc32_consumer<NCB, 1>(a, smem, lane, s0, S);
    else if (wave == 2) pc32_consumer<NCB, 2>(a, smem, lane, s0, S);
    else if (wave == 3) pc32_consumer<NCB, 3>(a, smem, lane, s0, S);
    else pc32_producer<NCB>(a, smem, lane, wave - 4, s0, S);
}

}  // namespace mirlsq
