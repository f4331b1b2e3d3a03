// jtj_fdp.h -- finite-difference rows -> J, J^T J, J^T y with register-staged producer waves (gfx950, f64).
//
// The m x 2n row-major panel of perturbed residuals Y[i][2j] = f(x + h e_j)_i, Y[i][2j+1] = f(x - h e_j)_i becomes the
// Jacobian (LS:1041-1047), which is written to J and contracted to J^T J (LS:1065) and J^T y (LS:1052) in the same pass.
// (Round 1's kernel for this job kept the bytes in flight in an LDS-DMA ring, 64 KB per CU, whose read side topped out near
// 2.9 TB/s: DESIGN_HISTORY.md.) Here the bytes in flight live in REGISTERS:
//
//   * a workgroup has 8 waves: 4 PRODUCERS and 4 CONSUMERS (the MFMA "roles" of jtj_kernel.h: the lower block triangle dealt
//     to four waves, 9 accumulator blocks each at n = 128);
//   * a stage is 32 rows; producer w owns rows 8w .. 8w+7 of every stage, i.e. 16n contiguous doubles of the panel,
//     which it reads with 2 NCB coalesced 16-byte loads per lane (the flat pair index 64 i + lane: row = f / n,
//     column = f % n): 64 VGPRs = 16 KB per wave, 64 KB per workgroup, 128 KB per CU in flight at n = 128 -- twice the
//     ring's. It forms J = (Y+ - Y-) (1 / twh) ONCE per element and writes the stage, in the plain row-major layout the
//     MFMA fragments are read from, into one of two LDS slots (ds_write_b64, lane-consecutive), then issues the loads of
//     the next stage, which fly while it waits at the barrier;
//     (a deeper register pipeline -- loads of stage t + 2 issued before stage t + 1 is consumed -- needs waits counted
//     across the loop back edge: the compiler's are maximally conservative there, and untracked inline-asm loads are
//     unsafe because the register allocator copies "defined" destination registers around while the data is in flight);
//   * one workgroup barrier per stage: after barrier t the consumers read stage t (ds_read_b64 fragments, MFMAs, J^T y)
//     while the producers convert stage t + 1 into the other slot; consumer roles 2 and 3 also write the J rows to
//     HBM from their fragments (128-byte segments), so the producers' vmcnt stream contains loads only.
//
// Three sources, one consumer side: the (+h, -h) pair panel (FD = true; the description above), J itself (FD = false: plain J^T J,
// nothing written back), and the m x n DIFFERENCE panel the caller's kernel has already subtracted (FD = false, DIFF = true:
// mir_lsq_gpu_options.fbRowMajorDiff -- the plain producer applies scal(1 / twh), the consumers write J): half the panel bytes.
//
// Slabs are laid out exactly like k_jtj's (jtj_kernel.h), so k_jtj_slab_reduce finishes the job. Rows past m are clamped on the
// load side and written as zeros to LDS.
#pragma once

#include "jtj_kernel.h"

namespace mirlsq {

constexpr int kJtjFdpThreads = 8 * kWave;

template <int NCB, bool FD = true> struct JtjFdpCfg {
    static constexpr int N = 16 * NCB;
    static constexpr int RP = !FD ? 8 : ((N % 64 == 0) ? 8 : 4);   // rows per producer wave and stage (FD: per-lane column
                                                                   // tables cost registers when n is not a multiple of 64)
    static constexpr int RS = 4 * RP;                      // rows per stage
    static constexpr int GPS = RS / 4;
    static constexpr int NI = FD ? RP * NCB / 4 : RP * NCB / 8;   // 16-byte loads per lane and stage
    static constexpr int SLOT_DOUBLES = RS * N + RS;       // the J stage, then its y values
    static constexpr int LDS_BYTES = 2 * SLOT_DOUBLES * 8;
    static constexpr int THREADS = 8 * kWave;
};

typedef double fdp_v2d __attribute__((ext_vector_type(2)));

// Producer wave w owns rows RP w .. RP w + RP - 1 of every stage: RP * 2n contiguous doubles of the panel, read with NI
// coalesced 16-byte loads per lane (flat pair index f = 64 i + lane: row = f / n, column = f % n). The loads of stage
// t + 1 are issued right after stage t has been converted and are in flight while the wave waits at the barrier for the
// consumers (64 VGPRs = 16 KB per wave, 128 KB per CU at n = 128); one buffer, so every wait the compiler inserts is a
// plain "everything I issued has landed" -- no counted waits across the loop back edge are needed.
template <int NCB>
__device__ __forceinline__ void fdp_producer(const JtjArgs<double>& a, double* smem, int lane, int w, size_t s0, size_t S)
{
    using C = JtjFdpCfg<NCB>;
    constexpr int n = C::N;                                // PADDED column count 16 NCB: the LDS stage layout
    constexpr int NI = C::NI;
    const size_t m = a.m;
    const int nr = a.n;                                    // the problem's n <= 16 NCB: row stride of Y (in pairs) and of J
    const fdp_v2d* __restrict__ Y = reinterpret_cast<const fdp_v2d*>(a.J);

    // n % 64 == 0: instruction i covers row (64 i) / n, columns (64 i) % n + lane -- the row is a compile-time constant and
    // only n / 64 distinct column sets exist; otherwise both are per-lane values. Padding columns (>= nr) behave like
    // collapsed intervals: they read a valid address and produce zeros.
    constexpr bool ALIGNED = n % 64 == 0;
    constexpr int KD = ALIGNED ? n / 64 : NI;              // distinct (column, 1 / twh) registers per lane
    int jrow_v[ALIGNED ? 1 : NI], jcol_v[KD];
    double inv[KD];
    bool zc[KD];
#pragma unroll
    for (int k = 0; k < KD; ++k) {
        const int f = 64 * k + lane;
        const int jc = ALIGNED ? f : f % n;
        const bool pad = jc >= nr;
        jcol_v[k] = pad ? 0 : jc;
        const double t = pad ? 0.0 : a.twh[jcol_v[k]];
        zc[k] = t == 0;                                    // collapsed interval: zero column (LS:1046)
        inv[k] = zc[k] ? 0.0 : 1.0 / t;                    // LS:1047
    }
    if constexpr (!ALIGNED) {
#pragma unroll
        for (int i = 0; i < NI; ++i) jrow_v[i] = (64 * i + lane) / n;
    }
    auto jrow = [&](int i) { if constexpr (ALIGNED) return (64 * i) / n; else return jrow_v[i]; };
    auto kd = [&](int i) { if constexpr (ALIGNED) return i % KD; else return i; };

    fdp_v2d b[NI];
    double yb = 0;
    auto issue = [&](size_t s) {
        const size_t row0 = (s0 + s) * C::RS + C::RP * (size_t)w;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            size_t row = row0 + jrow(i);
            row = row < m ? row : m - 1;
            b[i] = __builtin_nontemporal_load(&Y[row * (size_t)nr + jcol_v[kd(i)]]);   // read once: keep it out of the caches' way
        }
        size_t yr = row0 + (lane & (C::RP - 1));
        yr = yr < m ? yr : m - 1;
        yb = a.y[yr];
    };
    auto convert = [&](size_t s) {
        double* slot = smem + (s & 1) * C::SLOT_DOUBLES;
        const size_t row0 = (s0 + s) * C::RS + C::RP * (size_t)w;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            double d = b[i].x;                             // copy(mBuffer, Jj)       LS:1041
            d += -1.0 * b[i].y;                            // axpy(-1, mBuffer, Jj)   LS:1045
            double v = zc[kd(i)] ? 0.0 : d * inv[kd(i)];   // scal(1 / twh, Jj)       LS:1047
            v = (row0 + jrow(i) < m) ? v : 0.0;            // rows past m contribute nothing
            slot[w * C::RP * n + 64 * i + lane] = v;
        }
        if (lane < C::RP) slot[C::RS * n + C::RP * w + lane] = (row0 + lane < m) ? yb : 0.0;
    };

    if (S > 0) issue(0);
    for (size_t t = 0; t < S; ++t) {
        convert(t);
        if (t + 1 < S) issue(t + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS writes of stage t are done
        __builtin_amdgcn_s_barrier();                       // barrier t: stage t is complete for the consumers
    }
}

// Plain variant (FD = false): the source is J itself (m x n row-major, n even so that a row starts on a 16-byte
// boundary); a producer copies its RP rows of every stage into the LDS slot -- NI = NCB 16-byte loads per lane over the
// padded row (pairs past n read a valid address and are stored as zeros). Nothing is written back.
// DIFF: the source is the m x n DIFFERENCE panel D[i][j] = f(x + h e_j)_i - f(x - h e_j)_i (the caller's kernel has already
// done LS:1041 + LS:1045, mir_lsq_gpu_options.fbRowMajorDiff); the producer applies scal(1 / twh) (LS:1047, zero columns for
// collapsed intervals LS:1046) on the way to LDS and the consumers write the Jacobian rows out as in the pair-panel mode.
template <int NCB, bool DIFF = false>
__device__ __forceinline__ void fdp_producer_plain(const JtjArgs<double>& a, double* smem, int lane, int w, size_t s0, size_t S)
{
    using C = JtjFdpCfg<NCB, false>;
    constexpr int n = C::N;                                // padded
    constexpr int NI = C::NI;
    constexpr int HP = n / 2;                              // pairs per padded row
    const size_t m = a.m;
    const int nr = a.n;
    const size_t hr = (size_t)(nr / 2);                    // pairs per source row
    const fdp_v2d* __restrict__ Jp = reinterpret_cast<const fdp_v2d*>(a.J);

    int prow[NI], pcol[NI];                                // row inside the wave's RP rows, pair column (0 for padding)
    bool pad[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = 64 * i + lane;
        prow[i] = f / HP;
        const int jp = f % HP;
        pad[i] = 2 * jp >= nr;
        pcol[i] = pad[i] ? 0 : jp;
    }
    double inv0[DIFF ? NI : 1], inv1[DIFF ? NI : 1];       // 1 / twh of the pair's two columns (0: collapsed interval)
    if constexpr (DIFF) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const double t0 = a.twh[2 * pcol[i]], t1 = a.twh[2 * pcol[i] + 1];
            inv0[i] = t0 == 0 ? 0.0 : 1.0 / t0;
            inv1[i] = t1 == 0 ? 0.0 : 1.0 / t1;
        }
    }

    // TWO stages of loads in flight per producer wave (a ring of two register buffers): the loads of a stage are issued
    // unconditionally (past the end the last stage is read again), so the compiler can count them across the loop's back edge and
    // convert(t) waits for stage t only -- with `if (t + 1 < S) issue(t + 1)` it could not, every wait was vmcnt(0) and one stage
    // in flight was all a producer could have (the "deeper register pipeline" that rounds 2-3 gave up on).
    fdp_v2d b[2][NI];
    double yb[2] = {0, 0};
    auto issue = [&](size_t s, auto B) {
        constexpr int bb = decltype(B)::value;
        const size_t sc = s < S ? s : S - 1;
        const size_t row0 = (s0 + sc) * C::RS + C::RP * (size_t)w;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            size_t row = row0 + prow[i];
            row = row < m ? row : m - 1;
            if constexpr (DIFF) b[bb][i] = __builtin_nontemporal_load(&Jp[row * hr + pcol[i]]);   // the panel is read once
            else b[bb][i] = Jp[row * hr + pcol[i]];
        }
        size_t yr = row0 + (lane & (C::RP - 1));
        yr = yr < m ? yr : m - 1;
        yb[bb] = a.y[yr];
    };
    auto convert = [&](size_t s, auto B) {
        constexpr int bb = decltype(B)::value;
        double* slot = smem + (s & 1) * C::SLOT_DOUBLES;
        const size_t row0 = (s0 + s) * C::RS + C::RP * (size_t)w;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            fdp_v2d v = b[bb][i];
            if constexpr (DIFF) {
                v.x = inv0[i] == 0 ? 0.0 : v.x * inv0[i];      // scal(1 / twh, Jj), LS:1047 (LS:1046: zero column)
                v.y = inv1[i] == 0 ? 0.0 : v.y * inv1[i];
            }
            if (pad[i] || row0 + prow[i] >= m) v = fdp_v2d{0.0, 0.0};
            *reinterpret_cast<fdp_v2d*>(slot + w * C::RP * n + 2 * (64 * i + lane)) = v;
        }
        if (lane < C::RP) slot[C::RS * n + C::RP * w + lane] = (row0 + lane < m) ? yb[bb] : 0.0;
    };

    if (S > 0) {
        issue(0, IntC<0>{});
        issue(1, IntC<1>{});
        for (size_t t = 0; t < S; t += 2) {
            static_for<2>([&](auto U) {
                constexpr int u = decltype(U)::value;
                if (t + u < S) {
                    convert(t + u, U);
                    issue(t + u + 2, U);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS writes of the stage are done
                    __builtin_amdgcn_s_barrier();                        // the stage is complete for the consumers
                }
            });
        }
    }
}

// FLAT variant of the plain / difference-panel producer, for sources whose rows do not start on 16-byte boundaries -- ODD n
// (LS:911-926: the reference is generic in n) or an offset view of J. A wave's RP rows are RP n CONTIGUOUS doubles of the source
// (RP is even, so is the index of their first double): they are read as a flat array of pairs with 16-byte BUFFER loads --
// `buffer_load_dwordx4` needs 4-byte alignment only, and what lies past the end of the array (rows >= m, or the partner of the
// last double when m n is odd) reads as zero --, a pair may straddle a row end, so its two doubles go to LDS separately (two
// 8-byte stores at their own (row, column) positions). As many loads as the aligned variant; the padding columns of the LDS
// stage are zeroed once; 1 / twh of the difference panel comes from a small LDS table (n doubles behind the two slots).
template <int NCB, bool DIFF = false>
__device__ __forceinline__ void fdp_producer_plain_flat(const JtjArgs<double>& a, double* smem, int lane, int w, size_t s0, size_t S)
{
    using C = JtjFdpCfg<NCB, false>;
    constexpr int n = C::N;                                // padded
    constexpr int NI = C::NI;
    const size_t m = a.m;
    const int nr = a.n;
    double* invt = smem + 2 * C::SLOT_DOUBLES;             // DIFF: 1 / twh per column (0: collapsed interval, LS:1046)
    if constexpr (DIFF) {
        for (int c = lane; c < nr; c += kWave) { const double t = a.twh[c]; invt[c] = t == 0 ? 0.0 : 1.0 / t; }
    }
    // this workgroup's rows as a buffer: everything past the end of the array reads as zero
    const size_t rowb = s0 * C::RS < m ? s0 * C::RS : m;
    const size_t left = (m - rowb) * (size_t)nr * sizeof(double);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(a.J + rowb * (size_t)nr), 0, (int)(left < 0xffffffffull ? left : 0xffffffffull), 0x00020000);

    // DIFF: the scaled pairs ARE the rows of J in memory order (same row stride as the panel): the producer writes them back
    // itself, 1 KB contiguous per instruction -- the consumers' 128-byte row segments straddle cache lines whenever a row of J
    // does not start on a 128-byte boundary (n % 16 != 0: 0.52 ms at n = 126 against 0.43 at n = 128 for 1e6 rows)
    __amdgpu_buffer_rsrc_t rout = rsrc;
    if constexpr (DIFF) {
        if (a.Jout) rout = __builtin_amdgcn_make_buffer_rsrc(a.Jout + rowb * (size_t)nr, 0, (int)(left < 0xffffffffull ? left : 0xffffffffull), 0x00020000);
    }
    int off0[NI], off1[NI];                                // LDS offsets (row * n + column inside the wave's region); -1: no element
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e0 = 2 * (64 * i + lane), e1 = e0 + 1;
        off0[i] = e0 < C::RP * nr ? (e0 / nr) * n + e0 % nr : -1;
        off1[i] = e1 < C::RP * nr ? (e1 / nr) * n + e1 % nr : -1;
    }
    // the padding columns (nr .. n - 1) of this wave's rows, both slots: zero for the whole kernel
    for (int e = lane; e < C::RP * (n - nr); e += kWave) {
        const int r = e / (n - nr), c = nr + e % (n - nr);
        smem[w * C::RP * n + r * n + c] = 0.0;
        smem[C::SLOT_DOUBLES + w * C::RP * n + r * n + c] = 0.0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    typedef unsigned int fdp_u4 __attribute__((ext_vector_type(4)));
    fdp_u4 b[2][NI];
    double yb[2] = {0, 0};
    auto issue = [&](size_t s, auto B) {
        constexpr int bb = decltype(B)::value;
        const size_t sc = s < S ? s : S - 1;
        const size_t row0 = (s0 + sc) * C::RS + C::RP * (size_t)w;
        const unsigned base = (unsigned)((row0 - rowb) * (size_t)nr * sizeof(double));
#pragma unroll
        for (int i = 0; i < NI; ++i)
            b[bb][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + 16u * (unsigned)(64 * i + lane)), 0, DIFF ? 2 : 0);   // nt: the panel is read once
        size_t yr = row0 + (lane & (C::RP - 1));
        yr = yr < m ? yr : m - 1;
        yb[bb] = a.y[yr];
    };
    auto convert = [&](size_t s, auto B) {
        constexpr int bb = decltype(B)::value;
        double* slot = smem + (s & 1) * C::SLOT_DOUBLES + w * C::RP * n;
        const size_t row0 = (s0 + s) * C::RS + C::RP * (size_t)w;
        const unsigned base = (unsigned)((row0 - rowb) * (size_t)nr * sizeof(double));
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            double v0 = __hiloint2double((int)b[bb][i].y, (int)b[bb][i].x), v1 = __hiloint2double((int)b[bb][i].w, (int)b[bb][i].z);
            if constexpr (DIFF) {
                const double i0 = invt[off0[i] >= 0 ? off0[i] % n : 0], i1 = invt[off1[i] >= 0 ? off1[i] % n : 0];
                v0 = i0 == 0 ? 0.0 : v0 * i0;               // scal(1 / twh, Jj), LS:1047 (LS:1046: zero column)
                v1 = i1 == 0 ? 0.0 : v1 * i1;
                if (a.Jout && off0[i] >= 0) {               // rows >= m lie past the buffer's end: the store is dropped
                    const fdp_u4 o = {(unsigned)__double2loint(v0), (unsigned)__double2hiint(v0), (unsigned)__double2loint(v1), (unsigned)__double2hiint(v1)};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rout, (int)(base + 16u * (unsigned)(64 * i + lane)), 0, 2);
                }
            }
            if (off0[i] >= 0) slot[off0[i]] = v0;
            if (off1[i] >= 0) slot[off1[i]] = v1;
        }
        if (lane < C::RP) smem[(s & 1) * C::SLOT_DOUBLES + C::RS * n + C::RP * w + lane] = (row0 + lane < m) ? yb[bb] : 0.0;
    };
    if (S > 0) {
        issue(0, IntC<0>{});
        issue(1, IntC<1>{});
        for (size_t t = 0; t < S; t += 2) {
            static_for<2>([&](auto U) {
                constexpr int u = decltype(U)::value;
                if (t + u < S) {
                    convert(t + u, U);
                    issue(t + u + 2, U);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            });
        }
    }
}

template <int NCB, int ROLE, bool FD = true, bool DIFF = false, bool PRODUCER_WRITES_J = false>
__device__ __forceinline__ void fdp_consumer(const JtjArgs<double>& a, const double* smem, int lane, size_t s0, size_t S)
{
    using T = double;
    using Acc = typename Mma<T>::Acc;
    using C = JtjFdpCfg<NCB, FD>;
    constexpr int NACC = jtj_nacc<NCB>();
    constexpr int n = C::N;
    const int q = lane >> 4, p = lane & 15;
    const size_t m = a.m;

    Acc acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = Acc{0, 0, 0, 0};
    T jy[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) jy[c] = 0;

    for (size_t s = 0; s < S; ++s) {
        __builtin_amdgcn_s_barrier();                       // barrier s: the producers have written stage s
        const T* slot = smem + (s & 1) * C::SLOT_DOUBLES;
        const size_t row0 = (s0 + s) * C::RS;
        struct Grp { T v[NCB]; T y; };
        auto read = [&](int gi, Grp& g) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) g.v[c] = slot[(4 * gi + q) * n + 16 * c + p];
            g.y = 0;
            if constexpr (ROLE == 0) g.y = slot[C::RS * n + 4 * gi + q];
        };
        auto side = [&](int gi, const Grp& g) {
            if constexpr ((FD || DIFF) && ROLE >= 2 && !PRODUCER_WRITES_J) {
                // the Jacobian rows leave through roles 2 and 3 (column blocks c = ROLE (mod 2))
                const size_t row = row0 + 4 * gi + q;
                if (row < m && a.Jout) {                   // Jout == nullptr: the caller keeps the panel itself as J (unscaled)
                    T* wp = a.Jout + row * (size_t)a.n;    // the problem's n: row stride of J; padding columns stay in LDS
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        if (c % 2 == ROLE - 2) { if (16 * c + p < a.n) __builtin_nontemporal_store(g.v[c], &wp[16 * c + p]); }
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
        Grp ga, gb;
        read(0, ga);
        side(0, ga);
#pragma unroll
        for (int gi = 0; gi < C::GPS; gi += 2) {
            if (gi + 1 < C::GPS) read(gi + 1, gb);
            mfmas(ga);
            if (gi + 1 < C::GPS) {
                side(gi + 1, gb);
                if (gi + 2 < C::GPS) read(gi + 2, ga);
                mfmas(gb);
                if (gi + 2 < C::GPS) side(gi + 2, ga);
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

template <int NCB, bool FD = true, bool DIFF = false, bool ELEM = false>
__global__ __launch_bounds__(kJtjFdpThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_jtj_fdp(JtjArgs<double> a)
{
    static_assert(!(FD && DIFF), "DIFF uses the plain stage layout");
    static_assert(!(FD && ELEM), "the pair panel's rows always start on 16-byte boundaries");
    using C = JtjFdpCfg<NCB, FD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char fdp_smem[];
    double* smem = reinterpret_cast<double*>(fdp_smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const size_t Stot = (a.m + C::RS - 1) / C::RS;
    const size_t per = (Stot + gridDim.x - 1) / gridDim.x;
    const size_t s0 = (size_t)blockIdx.x * per < Stot ? (size_t)blockIdx.x * per : Stot;
    const size_t s1 = s0 + per < Stot ? s0 + per : Stot;
    const size_t S = s1 - s0;

    constexpr bool PW = DIFF && ELEM;                      // the flat producer of the difference panel writes J itself
    if (wave == 0) fdp_consumer<NCB, 0, FD, DIFF, PW>(a, smem, lane, s0, S);
    else if (wave == 1) fdp_consumer<NCB, 1, FD, DIFF, PW>(a, smem, lane, s0, S);
    else if (wave == 2) fdp_consumer<NCB, 2, FD, DIFF, PW>(a, smem, lane, s0, S);
    else if (wave == 3) fdp_consumer<NCB, 3, FD, DIFF, PW>(a, smem, lane, s0, S);
    else if constexpr (FD) fdp_producer<NCB>(a, smem, lane, wave - 4, s0, S);
    else if constexpr (ELEM) fdp_producer_plain_flat<NCB, DIFF>(a, smem, lane, wave - 4, s0, S);
    else fdp_producer_plain<NCB, DIFF>(a, smem, lane, wave - 4, s0, S);
}

}  // namespace mirlsq
