// jtj_fdp8.h -- finite-difference rows -> J, J^T J, J^T y for 128 < n <= 256 (f64, any n and m; compiled for N = 160, 192, 224,
// 256 = n rounded up to a multiple of 32, padding columns are zero columns): cfg 4's per-GPU shape.
//
// Same job as k_jtj_fdp (jtj_fdp.h): the m x 2n row-major panel of perturbed residuals Y[i][2j] = f(x + h e_j)_i,
// Y[i][2j+1] = f(x - h e_j)_i becomes the Jacobian (LS:1041-1047), which is written to J and contracted to J^T J
// (LS:1065) and J^T y (LS:1052) in the same pass -- 8 (3 m n + m) bytes instead of the 8 (4 m n + m) of k_fd_fill + k_jtj8.
//
// At n = 256 the lower triangle of J^T J is 136 MFMA blocks = 1088 accumulator VGPRs: eight waves ("roles", 17 blocks each)
// per workgroup, one workgroup per CU, 256 VGPRs per wave -- there is no room for separate producer waves as in k_jtj_fdp
// (12 waves would leave 170 VGPRs each). So EVERY wave is producer and consumer:
//   * a stage is 16 rows; wave w owns rows 2w, 2w + 1 of every stage as producer: 2 x 2n contiguous doubles of the panel,
//     read with n / 32 coalesced 16-byte loads per lane (one (+h, -h) pair each) into registers (32 VGPRs at n = 256;
//     64 KB per CU in flight);
//   * per stage t: (1) MFMA phase on stage t from LDS slot t & 1 -- while the loads of stage t + 1 are in flight --,
//     (2) convert stage t + 1: J = (Y+ - Y-) (1 / twh) once per element, to LDS slot (t + 1) & 1 AND straight to HBM from the
//     producer's registers (512 contiguous bytes per instruction), (3) issue the loads of stage t + 2, (4) ONE barrier;
//   * the LDS stage is row-major with a row stride of n + 16 doubles, so that the 4-row fragment reads of the MFMA operand
//     layout (lane (q, p) reads row 4 g + q, column 16 c + p) hit different bank halves for q = 0 / 1 (n itself is a multiple
//     of 32 doubles: all four sub-rows would share their banks);
//   * fragments are single-buffered (136 accumulators + 32 fragment + 32 load registers must fit 256): the two waves of a
//     SIMD cover each other's LDS latency;
//   * J^T y: role r accumulates the column blocks c with c * 8 / NCB == r (two each at n = 256).
// Slabs are laid out like k_jtj8's, so k_jtj_slab_reduce finishes the job. Rows past m are clamped on the load side and
// written as zeros to LDS (and not at all to J). Bound at m = 1e6, n = 256: HBM 6.2 GB at ~5 TB/s = 1.2 ms, MFMA 0.9 ms.
#pragma once

#include "jtj_fdp.h"

namespace mirlsq {

template <int NCB> struct JtjFdp8Cfg {
    static constexpr int N = 16 * NCB;
    static constexpr int ROLES = 8;
    static constexpr int RS = 16;                          // rows per stage
    static constexpr int RP = RS / ROLES;                  // rows per wave and stage (producer share)
    static constexpr int GPS = RS / 4;                     // 4-row groups per stage
    static constexpr int NI = RP * N / 64;                 // 16-byte loads per lane and stage
    static constexpr int LDJ = N + 16;                     // row stride of the LDS stage (doubles)
    static constexpr int SLOT_DOUBLES = RS * LDJ + RS;     // the J stage, then its y values
    static constexpr int LDS_BYTES = 2 * SLOT_DOUBLES * 8;
    static constexpr int THREADS = ROLES * kWave;
    static_assert(N % 32 == 0, "a wave's two rows must be a whole number of 64-pair loads");
};

// DIFF: a.J is the m x N row-major DIFFERENCE panel D[i][j] = f(x + h e_j)_i - f(x - h e_j)_i (mir_lsq_gpu_options.fbRowMajorDiff):
// a 16-byte load is two columns of one row, half as many loads per stage; the producer applies scal(1 / twh) only.
// PLAIN: a.J is J itself (m x n row-major, any n <= N and any m): the plain J^T J + J^T y of a given Jacobian (resynchronisation
// after a flush, analytic g) for the shapes the eight-wave ring does not take (n % 16 != 0, odd m). A wave's two rows of a stage
// are 2 n contiguous doubles, read as a flat array of pairs with 16-byte BUFFER loads (4-byte alignment suffices; past the end of
// the array reads as zero), each double stored to its own (row, column) of the LDS stage; nothing is written back.
template <int NCB, int ROLE, bool DIFF = false, bool PLAIN = false>
__device__ __forceinline__ void jtj_fdp8_body(const JtjArgs<double>& a, double* smem, int lane, size_t s0, size_t S)
{
    static_assert(!(DIFF && PLAIN), "one source at a time");
    using T = double;
    using Acc = typename Mma<T>::Acc;
    using C = JtjFdp8Cfg<NCB>;
    constexpr int NACC = jtj_nacc<NCB>();
    constexpr int N = C::N;
    constexpr int NI = DIFF ? C::NI / 2 : C::NI;
    constexpr int NPR = DIFF ? N / 2 : N;                  // 16-byte loads per panel row
    const int q = lane >> 4, p = lane & 15;
    const size_t m = a.m;
    const fdp_v2d* __restrict__ Y = reinterpret_cast<const fdp_v2d*>(a.J);   // m x n pairs (DIFF: m x N / 2 column pairs)
    // The problem's n may be smaller than the padded N = 16 NCB of this instantiation (pair panel only: any 128 < n <= 256 runs
    // the next even NCB; LS:911-926 is generic in n): nr is the row stride of the panel (in pairs) and of J; pairs of padding
    // columns read a valid address and behave like collapsed intervals (zeros in LDS, nothing written to J).
    const int nr = DIFF ? N : a.n;                         // (PLAIN: the row stride of J)

    // ---- producer side: load i of this lane is pair (row prow[i] of the wave's two rows, column pcol[i])
    //      (DIFF: columns 2 pcol[i], 2 pcol[i] + 1)
    int prow[NI], pcol[NI];
    T inv[NI], inv1[DIFF ? NI : 1];
    bool zc[NI];
    // PLAIN: pair-load i of this lane covers doubles e0 = 2 (64 i + lane), e0 + 1 of the wave's 2 n contiguous doubles
    constexpr int NIP = (N + 63) / 64;
    int off0[PLAIN ? NIP : 1], off1[PLAIN ? NIP : 1];      // LDS offsets inside the slot (-1: no element)
    typedef unsigned int fdp8_u4 __attribute__((ext_vector_type(4)));
    fdp8_u4 bp[PLAIN ? NIP : 1];
    const size_t rowb = s0 * C::RS < m ? s0 * C::RS : m;
    const size_t left = PLAIN ? (m - rowb) * (size_t)nr * sizeof(T) : 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(a.J + (PLAIN ? rowb * (size_t)nr : 0)), 0, (int)(left < 0xffffffffull ? left : 0xffffffffull), 0x00020000);
    if constexpr (PLAIN) {
#pragma unroll
        for (int i = 0; i < NIP; ++i) {
            const int e0 = 2 * (64 * i + lane), e1 = e0 + 1;
            off0[i] = e0 < C::RP * nr ? (C::RP * ROLE + e0 / nr) * C::LDJ + e0 % nr : -1;
            off1[i] = e1 < C::RP * nr ? (C::RP * ROLE + e1 / nr) * C::LDJ + e1 % nr : -1;
        }
        // the padding columns (nr .. N - 1) of this wave's rows, both slots: zero for the whole kernel
        for (int e = lane; e < C::RP * (N - nr); e += kWave) {
            const int o = (C::RP * ROLE + e / (N - nr)) * C::LDJ + nr + e % (N - nr);
            smem[o] = 0.0;
            smem[C::SLOT_DOUBLES + o] = 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < (PLAIN ? 0 : NI); ++i) {
        const int f = 64 * i + lane;
        prow[i] = f / NPR;
        pcol[i] = f % NPR;
        const bool padc = !DIFF && pcol[i] >= nr;
        const T t = padc ? T(0) : a.twh[DIFF ? 2 * pcol[i] : pcol[i]];
        if (padc) pcol[i] |= 0x10000;                      // bit 16: a padding column (address 0 of the row is read instead)
        zc[i] = t == 0;                                    // collapsed interval: zero column (LS:1046)
        inv[i] = zc[i] ? 0.0 : 1.0 / t;                    // LS:1047
        if constexpr (DIFF) {
            const T t1 = a.twh[2 * pcol[i] + 1];
            inv1[i] = t1 == 0 ? 0.0 : 1.0 / t1;
        }
    }
    fdp_v2d b[NI];
    T yb = 0;
    auto issue = [&](size_t t) {
        const size_t row0 = (s0 + t) * C::RS + C::RP * (size_t)ROLE;
        if constexpr (PLAIN) {
            const unsigned base = (unsigned)((row0 - rowb) * (size_t)nr * sizeof(T));
#pragma unroll
            for (int i = 0; i < NIP; ++i) bp[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + 16u * (unsigned)(64 * i + lane)), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < (PLAIN ? 0 : NI); ++i) {
            size_t row = row0 + prow[i];
            row = row < m ? row : m - 1;
            b[i] = __builtin_nontemporal_load(&Y[row * (size_t)(DIFF ? NPR : nr) + ((pcol[i] & 0x10000) ? 0 : pcol[i])]);   // the panel is read once
        }
        size_t yr = row0 + (lane & (C::RP - 1));
        yr = yr < m ? yr : m - 1;
        yb = a.y[yr];
    };
    auto convert = [&](size_t t) {
        T* slot = smem + (t & 1) * C::SLOT_DOUBLES;
        const size_t row0 = (s0 + t) * C::RS + C::RP * (size_t)ROLE;
        if constexpr (PLAIN) {
#pragma unroll
            for (int i = 0; i < NIP; ++i) {
                if (off0[i] >= 0) slot[off0[i]] = __hiloint2double((int)bp[i].y, (int)bp[i].x);
                if (off1[i] >= 0) slot[off1[i]] = __hiloint2double((int)bp[i].w, (int)bp[i].z);
            }
        }
#pragma unroll
        for (int i = 0; i < (PLAIN ? 0 : NI); ++i) {
            const size_t row = row0 + prow[i];
            const bool rok = row < m;
            if constexpr (DIFF) {
                fdp_v2d v;                                 // the caller did LS:1041 + 1045; scal(1 / twh, Jj), LS:1047
                v.x = (zc[i] || !rok) ? 0.0 : b[i].x * inv[i];
                v.y = (inv1[i] == 0 || !rok) ? 0.0 : b[i].y * inv1[i];
                *reinterpret_cast<fdp_v2d*>(slot + (C::RP * ROLE + prow[i]) * C::LDJ + 2 * pcol[i]) = v;
                if (rok && a.Jout) __builtin_nontemporal_store(v, reinterpret_cast<fdp_v2d*>(a.Jout + row * (size_t)N + 2 * pcol[i]));   // nullptr: the panel stays J
            } else {
                T d = b[i].x;                              // copy(mBuffer, Jj)       LS:1041
                d += -1.0 * b[i].y;                        // axpy(-1, mBuffer, Jj)   LS:1045
                T v = zc[i] ? 0.0 : d * inv[i];            // scal(1 / twh, Jj)       LS:1047
                v = rok ? v : 0.0;                         // rows past m contribute nothing
                slot[(C::RP * ROLE + prow[i]) * C::LDJ + (pcol[i] & 0xffff)] = v;
                if (rok && !(pcol[i] & 0x10000)) __builtin_nontemporal_store(v, &a.Jout[row * (size_t)nr + pcol[i]]);
            }
        }
        if (lane < C::RP) slot[C::RS * C::LDJ + C::RP * ROLE + lane] = (row0 + lane < m) ? yb : 0.0;
    };

    // ---- consumer side
    Acc acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = Acc{0, 0, 0, 0};
    T jy[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) jy[c] = 0;
    auto consume = [&](size_t t) {
        const T* slot = smem + (t & 1) * C::SLOT_DOUBLES;
#pragma unroll
        for (int gi = 0; gi < C::GPS; ++gi) {
            T v[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) v[c] = slot[(4 * gi + q) * C::LDJ + 16 * c + p];
            const T yv = slot[C::RS * C::LDJ + 4 * gi + q];
#pragma unroll
            for (int c = 0; c < NCB; ++c)
                if (c * C::ROLES / NCB == ROLE) jy[c] += v[c] * yv;                       // LS:1052
#pragma unroll
            for (int I = 0; I < NCB; ++I)
#pragma unroll
                for (int Jb = 0; Jb <= I; ++Jb)
                    if (jtj_owns<NCB, C::ROLES, ROLE>(I * (I + 1) / 2 + Jb))
                        acc[I * (I + 1) / 2 + Jb] = Mma<T>::mma(v[I], v[Jb], acc[I * (I + 1) / 2 + Jb]);   // LS:1065
        }
    };

    if (S > 0) {
        issue(0);
        convert(0);
        if (S > 1) issue(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // stage 0 is in LDS
    }
    for (size_t t = 0; t < S; ++t) {
        consume(t);                                         // the loads of stage t + 1 are in flight meanwhile
        if (t + 1 < S) {
            convert(t + 1);                                 // slot (t + 1) & 1: every wave left it before the last barrier
            if (t + 2 < S) issue(t + 2);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS writes of stage t + 1 are done
        __builtin_amdgcn_s_barrier();
    }

    T* dst = a.slabs + (size_t)blockIdx.x * jtj_slab_len<NCB>();
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        if (jtj_owns<NCB, C::ROLES, ROLE>(i)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(i * 4 + r) * kWave + lane] = acc[i][r];
        }
#pragma unroll
    for (int c = 0; c < NCB; ++c)
        if (c * C::ROLES / NCB == ROLE) {
            T s = jy[c];
            s += wave_shfl_xor(s, 16);
            s += wave_shfl_xor(s, 32);
            dst[(NACC * 4 + c) * kWave + lane] = s;
        }
}

template <int NCB, bool DIFF = false, bool PLAIN = false>
__global__ __launch_bounds__(JtjFdp8Cfg<NCB>::THREADS, 1) void k_jtj_fdp8(JtjArgs<double> a)
{
    using C = JtjFdp8Cfg<NCB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char fdp8_smem[];
    double* smem = reinterpret_cast<double*>(fdp8_smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t Stot = (a.m + C::RS - 1) / C::RS;
    const size_t per = (Stot + gridDim.x - 1) / gridDim.x;
    const size_t s0 = (size_t)blockIdx.x * per < Stot ? (size_t)blockIdx.x * per : Stot;
    const size_t s1 = s0 + per < Stot ? s0 + per : Stot;
    const size_t S = s1 - s0;
    switch (wave) {
    case 0: jtj_fdp8_body<NCB, 0, DIFF, PLAIN>(a, smem, lane, s0, S); break;
    case 1: jtj_fdp8_body<NCB, 1, DIFF, PLAIN>(a, smem, lane, s0, S); break;
    case 2: jtj_fdp8_body<NCB, 2, DIFF, PLAIN>(a, smem, lane, s0, S); break;
    case 3: jtj_fdp8_body<NCB, 3, DIFF, PLAIN>(a, smem, lane, s0, S); break;
    case 4: jtj_fdp8_body<NCB, 4, DIFF, PLAIN>(a, smem, lane, s0, S); break;
    case 5: jtj_fdp8_body<NCB, 5, DIFF, PLAIN>(a, smem, lane, s0, S); break;
    case 6: jtj_fdp8_body<NCB, 6, DIFF, PLAIN>(a, smem, lane, s0, S); break;
    default: jtj_fdp8_body<NCB, 7, DIFF, PLAIN>(a, smem, lane, s0, S); break;
    }
}

}  // namespace mirlsq
