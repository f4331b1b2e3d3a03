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

namespace mirlsq {

template <typename T>
struct JtjArgs {
    const T* J;        // m x n row-major
    T* Jout;           // BROYDEN: where updated rows are written (== J for in-place)
    const T* y;        // residual at the current point (length m)
    const T* y_old;    // BROYDEN: residual at the previous point (the reference's mBuffer after swap, LS:1136)
    const T* dx;       // BROYDEN: accepted step (length n)
    const T* dx_dot;   // BROYDEN: device scalar ||dx||^2 (LS:1002: d = 1 / deltaX_dot)
    T* slabs;          // gridDim.x slabs of jtj_slab_len<NCB>() elements
    size_t m;
    int n;
    const T* twh;      // finite-difference kernels (jtj_fdp.h, jtj_fdp8.h): interval widths xph - xmh (LS:1031); then J is the
                       // m x 2n row-major panel of perturbed residuals [f(x + h e_j), f(x - h e_j)]_j and Jout receives the Jacobian
};

constexpr int kJtjWaves = 4;   // waves per workgroup

template <int NCB> __host__ __device__ constexpr int jtj_nacc() { return NCB * (NCB + 1) / 2; }
// slab: NACC blocks x 4 registers x 64 lanes, then NCB x 64 lanes of J^T y partials
template <int NCB> __host__ __device__ constexpr int jtj_slab_len() { return (jtj_nacc<NCB>() * 4 + NCB) * kWave; }
// The accumulator blocks are split over 1, 2 or 4 "roles" (waves that walk the same rows) so that
// one wave keeps at most 96 accumulator VGPRs (regs_per_block = 4 for f32, 8 for f64).
__host__ __device__ constexpr int jtj_roles_rt(int ncb, int regs_per_block)
{
    const int regs = ncb * (ncb + 1) / 2 * regs_per_block;
    return regs <= 96 ? 1 : (regs <= 192 ? 2 : 4);
}
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

    struct Frag { T v[NCB]; T y, yo; };

    auto load = [&](size_t g, Frag& f) {
        const size_t row = 4 * g + q;
        const bool rok = row < m;
        const size_t rc = rok ? row : m - 1;
        const T* rp = a.J + rc * (size_t)n;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const T t = rp[coff[c]];
            f.v[c] = (rok && cok[c]) ? t : T(0);
        }
        const T t = a.y[rc];
        f.y = rok ? t : T(0);
        f.yo = 0;
        if constexpr (BROYDEN) { const T t2 = a.y_old[rc]; f.yo = rok ? t2 : T(0); }
    };

    auto compute = [&](size_t g, Frag& f) {
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

    // software pipeline, two named fragment sets (no register copies)
    Frag fa, fb;
    size_t g = g0;
    if (g < g1) load(g, fa);
    while (g < g1) {
        if (g + 1 < g1) load(g + 1, fb);
        compute(g, fa);
        ++g;
        if (g >= g1) break;
        if (g + 1 < g1) load(g + 1, fa);
        compute(g, fb);
        ++g;
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

// =========================================================================================
// v2: the same fused pass with an LDS-DMA ring (f64, n = 16 NCB, m even).
//
// v1 above keeps only one 4-row group per wave in flight, so a CU has a few KB of J outstanding and
// the kernel runs at HBM *latency* (measured 1.4 TB/s algorithmic at m = 1e6, n = 128). Here the
// four waves of a workgroup ("roles", each owning a quarter of the accumulator blocks and walking the
// same rows) stream J through a ring of NS LDS slots with `global_load_lds_dwordx4`: no VGPRs are
// spent on data in flight, up to D stages (~60 KB) per workgroup are outstanding, each wave issues
// its share of a stage's 1 KB DMA instructions, waits for its own with a COUNTED `s_waitcnt
// vmcnt(N)` (N = D x the VMEM operations the wave issues per stage, DMA + Broyden stores) and one
// raw `s_barrier` per stage publishes the stage to the other waves. The MFMA fragments are read
// from the slot with ds_read_b64: lane (q, p) reads row 4 g + q, column 16 c + p -> exactly the
// A/B operand layout. y / y_old ride in two small rings of their own (one 1 KB DMA per 128 rows),
// so the main loop contains no VGPR-destination global load at all.
// =========================================================================================
typedef __attribute__((address_space(3))) void* jtj_lds_ptr;
typedef const __attribute__((address_space(1))) void* jtj_gbl_ptr;

template <int NCB, bool BROYDEN> struct Jtj2Cfg {
    // rows per stage: ~16 KB stages so that one barrier is amortised over 4+ row groups (a probe on
    // MI355X: 4 KB stages 0.33 ms, 16 KB stages 0.27 ms for the n = 128 MFMA work; scripts/probes).
    static constexpr int RS = NCB <= 4 ? 32 : (NCB == 5 ? 24 : (NCB == 6 ? 20 : 16));
    static constexpr int GPS = RS / 4;                                                     // 4-row groups per stage
    static constexpr int IPS = RS * NCB / 8;                                               // 1 KB DMA instructions per stage
    static constexpr int SLOT_BYTES = IPS * 1024;
    // DMA instructions the busier of the two loading waves issues per stage (waves 0, 1 issue every
    // DMA and never store; waves 2, 3 do every Broyden write-back and issue no DMA: a counted vmcnt
    // wait is only reliable over operations of one kind -- mixing stores into the count made the
    // DMA wait pass early now and then, a run-to-run nondeterminism caught by scripts/diag_determinism.py)
    static constexpr int MAX_OPS = (IPS + 1) / 2;
    static constexpr int D0 = 60 / MAX_OPS;                                                // vmcnt is 6 bits
    static constexpr int D1 = (64 * 1024) / SLOT_BYTES - 2;                                // 64 KB ring
    static constexpr int D2 = D0 < D1 ? D0 : D1;
    static constexpr int D = D2 < 1 ? 1 : (D2 > 15 ? 15 : D2);                             // stages in flight
    static constexpr int NS = D + 2;                                                       // ring slots
    static constexpr int RING_BYTES = NS * SLOT_BYTES;
    // y / y_old travel through their own small rings: one 1 KB DMA instruction = one chunk of 128 rows
    static constexpr int YNS = 4;                   // y ring slots (chunks c-1 .. c+2 may be live)
    static constexpr int Y_OFF = RING_BYTES;
    static constexpr int YO_OFF = RING_BYTES + YNS * 1024;
    static constexpr int LDS_BYTES = RING_BYTES + 2 * YNS * 1024;
};
constexpr int kJtj2Threads = 4 * kWave;

template <int NCB, bool BROYDEN, int ROLE>
__device__ __forceinline__ void jtj2_body(const JtjArgs<double>& a, unsigned char* smem, int lane, size_t s0, size_t S)
{
    using T = double;
    using Acc = typename Mma<T>::Acc;
    using C = Jtj2Cfg<NCB, BROYDEN>;
    constexpr int NACC = jtj_nacc<NCB>();
    constexpr int n = 16 * NCB;
    constexpr int RW = n;                                    // doubles per source row
    // this wave's share of a stage: waves 0, 1 issue the DMA instructions (ROLE, ROLE + 2, ...) and never
    // store; waves 2, 3 write back the Broyden-updated column blocks c = ROLE (mod 2) and never load
    constexpr bool LOADER = ROLE < 2;
    constexpr int MYI = LOADER ? (C::IPS + 1 - ROLE) / 2 : 0;
    constexpr int OPS = MYI;
    const int q = lane >> 4, p = lane & 15;
    const size_t m = a.m;

    const unsigned char* Jb = reinterpret_cast<const unsigned char*>(a.J);
    const size_t total = m * (size_t)RW * sizeof(T);
    auto issue = [&](size_t s) {
        const size_t base = (s0 + s) * (size_t)C::RS * RW * sizeof(T);
        unsigned char* slot = smem + (s % C::NS) * C::SLOT_BYTES;
#pragma unroll
        for (int k = 0; k < MYI; ++k) {
            const int ins = ROLE + 2 * k;
            size_t off = base + (size_t)(ins * 64 + lane) * 16;
            if (off + 16 > total) off = base;            // rows past m: any valid bytes (masked by the consumers)
            __builtin_amdgcn_global_load_lds((jtj_gbl_ptr)(Jb + off), (jtj_lds_ptr)(slot + ins * 1024), 16, 0, 2 /* nt: J is swept once per pass */);
        }
    };
    // y / y_old chunks (128 rows each) are issued by role 1, two chunks ahead, in the same in-order queue
    const unsigned char* yb = reinterpret_cast<const unsigned char*>(a.y);
    const unsigned char* yob = reinterpret_cast<const unsigned char*>(a.y_old);
    const size_t ytotal = m * sizeof(T);
    const size_t nchunks = (S * C::RS + 127) / 128;
    size_t next_chunk = 0;
    auto issue_y = [&](size_t c) {
        size_t off = ((s0 * C::RS) + c * 128) * sizeof(T) + (size_t)lane * 16;
        if (off + 16 > ytotal) off = 0;
        __builtin_amdgcn_global_load_lds((jtj_gbl_ptr)(yb + off), (jtj_lds_ptr)(smem + C::Y_OFF + (c % C::YNS) * 1024), 16, 0, 0);
        if constexpr (BROYDEN)
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
        if constexpr (BROYDEN) dxr[c] = a.dx[16 * c + p];
    }
    T neg_d = 0;
    if constexpr (BROYDEN) neg_d = -(T(1) / *a.dx_dot);
    // the loads above must have retired before the counted waits below start counting
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if constexpr (ROLE == 1) {
        while (next_chunk < 3 && next_chunk < nchunks) issue_y(next_chunk++);
    }
    constexpr int L = C::D;
    const size_t pre = S < (size_t)L ? S : (size_t)L;
    for (size_t s = 0; s < pre; ++s) issue(s);

    for (size_t s = 0; s < S; ++s) {
        if constexpr (ROLE == 1) {
            // keep the chunk holding this stage's first row plus two more in flight / resident
            while (next_chunk <= (s * C::RS) / 128 + 2 && next_chunk < nchunks) issue_y(next_chunk++);
        }
        if (s + C::D < S) {
            issue(s + C::D);
            // my DMA of stage s (and everything older) has landed once at most D * OPS younger ops remain
            if constexpr (OPS > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::D * OPS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                       // stage s is complete in LDS for every wave
        const T* slot = reinterpret_cast<const T*>(smem + (s % C::NS) * C::SLOT_BYTES);
        const T* yring = reinterpret_cast<const T*>(smem + C::Y_OFF);
        const T* yoring = reinterpret_cast<const T*>(smem + C::YO_OFF);
        const size_t row0 = (s0 + s) * C::RS;               // first global row of the stage

        // One group = 4 rows: fragments v[NCB] (+ y, y_old). Software pipeline inside the stage:
        //   read(g + 1) is issued before the MFMAs of g, and prepare(g + 1) -- masking, Broyden update,
        //   write-back, J^T y -- is independent VALU work the scheduler can slot between those MFMAs.
        struct Grp { T v[NCB]; T y, yo; };
        auto read = [&](int gi, Grp& g) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                g.v[c] = slot[(4 * gi + q) * n + 16 * c + p];
            }
            const size_t lr = s * C::RS + 4 * gi + q;       // row index local to this workgroup
            const int yidx = (int)((lr >> 7) % C::YNS) * 128 + (int)(lr & 127);
            g.y = yring[yidx];
            g.yo = 0;
            if constexpr (BROYDEN) g.yo = yoring[yidx];
        };
        auto prepare = [&](int gi, Grp& g, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            const size_t row = row0 + 4 * gi + q;
            const bool rok = FULL ? true : row < m;
            if constexpr (!FULL) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) g.v[c] = rok ? g.v[c] : T(0);
                g.y = rok ? g.y : T(0);
                g.yo = rok ? g.yo : T(0);
            }
            if constexpr (BROYDEN) {
                T part = 0;
#pragma unroll
                for (int c = 0; c < NCB; ++c) part += g.v[c] * dxr[c];
                part = sum16(part);
                const T t = (g.yo - g.y) + part;          // LS:1003-1004
                const T u = neg_d * t;                    // LS:1005
                T* wp = a.Jout + (rok ? row : m - 1) * (size_t)n;
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    g.v[c] = g.v[c] + u * dxr[c];         // LS:1006
                    if constexpr (!LOADER) { if (c % 2 == ROLE - 2) { if (rok) wp[16 * c + p] = g.v[c]; } }
                }
            }
            if constexpr (ROLE == 0) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) jy[c] += g.v[c] * g.y;     // LS:1052
            }
        };
        auto mfmas = [&](const Grp& g) {
#pragma unroll
            for (int I = 0; I < NCB; ++I)
#pragma unroll
                for (int Jb2 = 0; Jb2 <= I; ++Jb2)
                    if (jtj_owns<NCB, 4, ROLE>(I * (I + 1) / 2 + Jb2))
                        acc[I * (I + 1) / 2 + Jb2] = Mma<T>::mma(g.v[I], g.v[Jb2], acc[I * (I + 1) / 2 + Jb2]);   // LS:1065
        };
        auto stage = [&](auto full_tag) {
            Grp ga, gb;
            read(0, ga);
            prepare(0, ga, full_tag);
#pragma unroll
            for (int gi = 0; gi < C::GPS; gi += 2) {
                if (gi + 1 < C::GPS) read(gi + 1, gb);
                mfmas(ga);
                if (gi + 1 < C::GPS) {
                    prepare(gi + 1, gb, full_tag);
                    if (gi + 2 < C::GPS) read(gi + 2, ga);
                    mfmas(gb);
                    if (gi + 2 < C::GPS) prepare(gi + 2, ga, full_tag);
                }
            }
        };
        if (row0 + C::RS <= m) stage(std::true_type{}); else stage(std::false_type{});
    }

    // every wave owns a disjoint part of the workgroup's slab: no LDS reduction needed
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

template <int NCB, bool BROYDEN>
__global__ __launch_bounds__(kJtj2Threads, 2) void k_jtj2(JtjArgs<double> a)
{
    using C = Jtj2Cfg<NCB, BROYDEN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // contiguous range of stages for this workgroup
    const size_t Stot = (a.m + C::RS - 1) / C::RS;
    const size_t per = (Stot + gridDim.x - 1) / gridDim.x;
    const size_t s0 = (size_t)blockIdx.x * per < Stot ? (size_t)blockIdx.x * per : Stot;
    const size_t s1 = s0 + per < Stot ? s0 + per : Stot;
    const size_t S = s1 - s0;

    if (wave == 0) jtj2_body<NCB, BROYDEN, 0>(a, smem2, lane, s0, S);
    else if (wave == 1) jtj2_body<NCB, BROYDEN, 1>(a, smem2, lane, s0, S);
    else if (wave == 2) jtj2_body<NCB, BROYDEN, 2>(a, smem2, lane, s0, S);
    else jtj2_body<NCB, BROYDEN, 3>(a, smem2, lane, s0, S);
}

}  // namespace mirlsq
