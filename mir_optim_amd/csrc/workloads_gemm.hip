// workloads_gemm.hip -- the batched tanh-linear residual of the synthetic workloads as a tall GEMM on the f64 matrix cores
// (caller side of the boundary, like workloads.hip: the role of the user's batched f, mir_lsq_batched_function_d).
#include "workloads_gemm.h"
#include "workloads_device.h"

namespace {

// ---- batched tanh-linear residuals: Y[p][i] = tanh(a_i . X[p]) - b_i for p < P points in ONE
//      sweep over A (the finite-difference Jacobian evaluates its 2n perturbed points together).
//      It is a tall GEMM A[m x n] X^T[n x P] on v_mfma_f64_16x16x4_f64:
//        * workgroup = 16 waves; wave w owns points [16 w, 16 w + 16) of a 256-point chunk and keeps
//          their X fragments (B operand, NK k-steps) in registers for the whole sweep;
//        * the workgroup streams 16-row tiles of A through a double-buffered, padded LDS image
//          (row pitch n + 2 doubles -> the 16-row x 4-column A-operand read is bank-conflict free);
//          every byte of A is read from HBM once per 256 points;
//        * epilogue: tanh - b on the accumulator, stored along the row index (4 x 32-byte runs per
//          point and tile, merged into full lines in L2).
template <int NK>
__global__ __launch_bounds__(1024) void k_tanh_linear_batched(const double* __restrict__ A, const double* __restrict__ b,
                                                               const double* __restrict__ X, double* __restrict__ Y,
                                                               size_t m, int n, int P)
{
    using Acc = __attribute__((ext_vector_type(4))) double;
    constexpr int NPAD = 4 * NK;            // padded column count
    constexpr int PITCH = NPAD + 2;         // doubles
    __shared__ __attribute__((aligned(16))) double tile[2][16 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    // zero the padding columns once (never overwritten afterwards)
    for (int idx = tid; idx < 2 * 16 * PITCH; idx += 1024) (&tile[0][0])[idx] = 0.0;
    __syncthreads();

    // piece -> (row, 16-byte column pair) of a 16-row tile; tile rows are contiguous in memory
    const int ppr = n / 2;                  // 16-byte pieces per row (n even)
    const bool has_piece = tid < 16 * ppr;
    const int prow = has_piece ? tid / ppr : 0, pcol = has_piece ? tid % ppr : 0;
    const size_t ntiles = (m + 15) / 16;

    for (int pbase = 0; pbase < P; pbase += 256) {
        const int pl = pbase + wave * 16 + fr;                  // this lane's point
        const bool pok = pl < P;
        const bool wave_active = pbase + wave * 16 < P;         // wave-uniform
        double xf[NK];
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            const int col = 4 * s + fq;
            xf[s] = (pok && col < n) ? X[(size_t)pl * n + col] : 0.0;
        }
        double2 stage = make_double2(0.0, 0.0);
        auto gload = [&](size_t t) {
            const size_t row = t * 16 + prow;
            if (has_piece && row < m) stage = *reinterpret_cast<const double2*>(A + row * (size_t)n + 2 * pcol);
            else stage = make_double2(0.0, 0.0);
        };
        auto lstore = [&](int buf) {
            if (has_piece) *reinterpret_cast<double2*>(&tile[buf][prow * PITCH + 2 * pcol]) = stage;
        };
        size_t t = blockIdx.x;
        if (t < ntiles) { gload(t); lstore(0); }
        if (t + gridDim.x < ntiles) gload(t + gridDim.x);
        __syncthreads();
        int buf = 0;
        // software pipeline over tiles: the tanh / store epilogue of tile t - 1 is independent VALU work placed in
        // the same basic block as the (dependent, latency-bound) MFMA chain of tile t, so the scheduler overlaps them
        Acc prev = {0.0, 0.0, 0.0, 0.0};
        size_t tprev = ntiles;                                  // "no previous tile"
        auto epilogue = [&](const Acc& acc, size_t tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t row = tt * 16 + fq + 4 * r;
                if (pok && tt < ntiles && row < m) Y[(size_t)pl * m + row] = dtanh(acc[r]) - b[row];
            }
        };
        for (; t < ntiles; t += gridDim.x) {
            if (wave_active) {                                  // waves whose 16 points are all >= P only help staging
                Acc acc = {0.0, 0.0, 0.0, 0.0};
                const double* tp = &tile[buf][fr * PITCH + fq];
#pragma unroll
                for (int s = 0; s < NK; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tp[4 * s], xf[s], acc, 0, 0, 0);
                epilogue(prev, tprev);                          // D: col = lane & 15 = point, row = (lane >> 4) + 4 r
                prev = acc;
                tprev = t;
            }
            // stage tile t + grid into the other buffer (its last readers passed the previous barrier)
            if (t + gridDim.x < ntiles) lstore(buf ^ 1);
            if (t + 2 * (size_t)gridDim.x < ntiles) gload(t + 2 * (size_t)gridDim.x);
            __syncthreads();
            buf ^= 1;
        }
        if (wave_active) epilogue(prev, tprev);
        __syncthreads();
    }
}


// ---- batched residuals, LDS-DMA version (n = 32, 64, 128): same GEMM, restructured like the solver's J^T J kernel.
//      The first version above stages A through VGPRs and its waves wait, every 16-row tile, on a vmcnt that also
//      covers their own scattered Y stores (measured 2.1 ms for m = 1e6, n = p = 128: 15 TFLOP/s). Here
//        * waves 0..7 compute (wave w owns points [16 w, 16 w + 16) of a 128-point chunk) and issue NO loads in the
//          sweep: their Y stores are fire-and-forget;
//        * waves 8, 9 only issue `global_load_lds_dwordx4` DMA into a ring of NS stages of 32 rows (D stages in flight,
//          counted vmcnt over DMA only, one raw s_barrier per stage); b rides in a small ring of its own;
//        * LDS layout: row-major without padding, 16-byte pieces XOR-swizzled with the row (the DMA picks the SOURCE
//          row and piece per lane, the LDS side of a DMA is always lane x 16 B). The K index of the MFMA chain is
//          permuted so that one ds_read_b128 (4 LDS cycles, conflict-free with this swizzle) is the A operand of two
//          k-steps; ds_read2_b64, which the compiler forms from paired 8-byte reads, costs 8 cycles and 2-way conflicts;
//        * LDS row i of a tile holds memory row rho(i), so that a lane's four accumulator rows are two adjacent row
//          pairs: the epilogue stores 16-byte pairs, 64 contiguous bytes per point and instruction;
//        * two independent accumulator chains (two 16-row tiles of the stage) per wave keep the MFMA pipe full; the two
//          compute waves of a SIMD run out of phase (MFMA-then-epilogue vs epilogue-then-MFMA).
typedef __attribute__((address_space(3))) void* wl_lds_ptr;
typedef const __attribute__((address_space(1))) void* wl_gbl_ptr;

// MFMA row i of a 16-row tile holds memory row rho(i): D row fq + 4 r  <->  memory row (r >> 1) * 8 + 2 fq + (r & 1)
__device__ __forceinline__ constexpr int tlb_rho(int i) { return ((i >> 2) >> 1) * 8 + 2 * (i & 3) + ((i >> 2) & 1); }
// 16-byte piece p of LDS row i is stored at piece position p ^ sigma(i): with it the 4 x 16-lane groups of a
// ds_read_b128 A-operand fetch ({0-3, 12-15, 20-27}, ...) each touch 16 distinct 16-byte bank slots
__device__ __forceinline__ constexpr int tlb_sigma(int i) { return (i + 12) & 15; }

template <int NK> struct TlbCfg {
    static constexpr int N = 4 * NK;
    static constexpr int PPR = N / 2;                       // 16-byte pieces per row
    // n <= 128: 8 compute waves x 16 points, two 16-row tiles per stage (two MFMA chains per wave), 168 VGPRs;
    // n = 256: the X fragments alone are 128 VGPRs -> 8 waves per workgroup (256 VGPRs each), one tile per stage with the
    // K range split over two accumulator chains
    static constexpr int TILES = NK <= 32 ? 2 : 1;
    // SELF (n = 256): no loader waves -- eight compute waves (two per SIMD, as at n <= 128) each issue an eighth of a stage's DMA
    // and wait for it with a count over the YOUNGER DMA only (see tlb_compute); with two loader waves among eight, two SIMDs
    // carried one compute wave and the MFMA pipe could not pass 0.75 (measured 0.59, A swept six times for 512 points)
    static constexpr bool SELF = NK > 32;
    static constexpr int COMPUTE_WAVES = 8, LOADER_WAVES = SELF ? 0 : 2;
    static constexpr int CHUNK = 16 * COMPUTE_WAVES;        // points per sweep over A
    static constexpr int ROWS = 16 * TILES;                 // rows per stage
    static constexpr int STAGE_BYTES = ROWS * N * 8;
    static constexpr int IPS = STAGE_BYTES / 1024;          // 1 KB DMA instructions per stage
    static constexpr int NS = 4, D = 2, NSB = 8;
    static constexpr int B_OFF = NS * STAGE_BYTES;
    static constexpr int LDS_BYTES = B_OFF + NSB * 256;
    static constexpr int THREADS = 64 * (COMPUTE_WAVES + LOADER_WAVES);
};

template <int NK, int LOADER>
__device__ __forceinline__ void tlb_loader(const double* __restrict__ A, const double* __restrict__ b, size_t m,
                                           unsigned char* smem, int lane, size_t S, size_t F)
{
    using C = TlbCfg<NK>;
    constexpr int MYI = C::IPS / 2;
    constexpr int OPS = MYI + (LOADER == 1 ? 1 : 0);
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(A);
    const unsigned char* bb = reinterpret_cast<const unsigned char*>(b);
    auto issue = [&](size_t f) {
        const size_t stage = blockIdx.x + (f % S) * (size_t)gridDim.x;
        const size_t row0 = stage * C::ROWS;
        unsigned char* slot = smem + (f % C::NS) * C::STAGE_BYTES;
#pragma unroll
        for (int k = 0; k < MYI; ++k) {
            const int ins = LOADER + 2 * k;
            const int g = ins * 64 + lane;                      // 16-byte piece of the stage image this lane fills
            const int R = g / C::PPR, sp = g % C::PPR;          // LDS row within the stage = 16 tile + MFMA row
            size_t row = row0 + (R & 16) + tlb_rho(R & 15);     // the memory row that feeds it
            row = row < m ? row : m - 1;                        // rows past m: valid bytes, never stored
            const int piece = sp ^ tlb_sigma(R & 15);
            __builtin_amdgcn_global_load_lds((wl_gbl_ptr)(Ab + (row * C::N + 2 * piece) * 8),
                                             (wl_lds_ptr)(slot + ins * 1024), 16, 0, 2 /* nt */);
        }
        if constexpr (LOADER == 1) {
            size_t row = row0 + (lane >> 1);
            row = row < m ? row : m - 1;
            __builtin_amdgcn_global_load_lds((wl_gbl_ptr)(bb + row * 8 + (lane & 1) * 4),
                                             (wl_lds_ptr)(smem + C::B_OFF + (f % C::NSB) * 256), 4, 0, 0);
        }
    };
    const size_t pre = F < (size_t)C::D ? F : (size_t)C::D;
    for (size_t f = 0; f < pre; ++f) issue(f);
    for (size_t f = 0; f < F; ++f) {
        if (f + C::D < F) {
            issue(f + C::D);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::D * OPS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
}

// RM: Y is m x P row-major (point k's residual of row i at Y[i P + k]; 16 lanes of a row = 128 contiguous bytes) --
// the layout the solver's fused finite-difference J^T J kernel consumes (mir_lsq_gpu_options.fbRowMajor)
// DIFF (with RM): points come in (+h, -h) pairs (2 j, 2 j + 1) -- adjacent lanes -- and Y is the m x P/2 row-major DIFFERENCE
// panel D[i][j] = y_{2j}[i] - y_{2j+1}[i] (mir_lsq_gpu_options.fbRowMajorDiff): the even lane of a pair subtracts its
// neighbour's residual (a DPP move) and stores; half the panel bytes of RM
template <int NK, bool ALIGNED, int GROUP, bool RM = false, bool DIFF = false>
__device__ __forceinline__ void tlb_compute(const double* __restrict__ A, const double* __restrict__ b,
                                            const double* __restrict__ X, double* __restrict__ Y, size_t m, int P,
                                            unsigned char* smem, int lane, int wave, size_t S, int nchunks)
{
    using C = TlbCfg<NK>;
    using Acc = __attribute__((ext_vector_type(4))) double;
    const int fr = lane & 15, fq = lane >> 4;
    // k-step s = 2 j + e of the MFMA chain multiplies column 8 j + 2 fq + e: one ds_read_b128 (piece 4 j + fq of LDS
    // row fr) feeds two k-steps, and X is read as the matching column pairs
    const int v = fq ^ tlb_sigma(fr);
    int laddr[4];                                               // byte offsets of j = 0..3 (mod 4) in a stage
#pragma unroll
    for (int k = 0; k < 4; ++k) laddr[k] = fr * C::N * 8 + ((4 * k) ^ v) * 16;

    // SELF: this wave's share of the DMA (instructions wave, wave + 8, ... of a stage; wave 0 also b). Loads return in issue
    // order among themselves, stores do not keep order with loads: a count of the YOUNGER LOADS ONLY is a safe wait for a
    // stage (if one of its loads were pending, so would be every younger one: more than the count) -- pending stores can only
    // make it wait longer, and the wait sits before the epilogue's stores, behind a stage's worth of matrix-core work.
    constexpr int MYI = C::SELF ? C::IPS / C::COMPUTE_WAVES : 1;
    [[maybe_unused]] int roff[MYI], soff[MYI];
    [[maybe_unused]] unsigned is_next = 0, slot_next = 0, bslot_next = 0;          // stage (of S), ring slot, b slot of the next issue
    [[maybe_unused]] const size_t Ftot = S * (size_t)nchunks;
    [[maybe_unused]] size_t fissued = 0;
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(A);
    const unsigned char* bb = reinterpret_cast<const unsigned char*>(b);
    if constexpr (C::SELF) {
#pragma unroll
        for (int k = 0; k < MYI; ++k) {
            const int g = (wave + C::COMPUTE_WAVES * k) * 64 + lane;
            const int R = g / C::PPR, sp = g % C::PPR;
            roff[k] = (R & 16) + tlb_rho(R & 15);
            soff[k] = 2 * (sp ^ tlb_sigma(R & 15)) * 8;
        }
    }
    auto issue = [&]() {
        const size_t row0 = (blockIdx.x + (size_t)is_next * gridDim.x) * C::ROWS;
        unsigned char* slot = smem + slot_next * C::STAGE_BYTES;
#pragma unroll
        for (int k = 0; k < MYI; ++k) {
            size_t row = row0 + roff[k];
            row = row < m ? row : m - 1;
            __builtin_amdgcn_global_load_lds((wl_gbl_ptr)(Ab + row * (C::N * 8) + soff[k]),
                                             (wl_lds_ptr)(slot + (wave + C::COMPUTE_WAVES * k) * 1024), 16, 0, 2 /* nt */);
        }
        if (wave == 0) {
            size_t row = row0 + (lane >> 1);
            row = row < m ? row : m - 1;
            __builtin_amdgcn_global_load_lds((wl_gbl_ptr)(bb + row * 8 + (lane & 1) * 4),
                                             (wl_lds_ptr)(smem + C::B_OFF + bslot_next * 256), 4, 0, 0);
        }
        is_next = is_next + 1 == (unsigned)S ? 0 : is_next + 1;
        slot_next = (slot_next + 1) % C::NS;
        bslot_next = (bslot_next + 1) % C::NSB;
        ++fissued;
    };
    // own share of stage ff complete (stages ff + 1 .. ff + D issued behind it)
    auto advance = [&]() {
        if constexpr (C::SELF) {
            if (fissued < Ftot) {
                issue();
                if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::D * (MYI + 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::D * MYI) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    };
    if constexpr (C::SELF) {
        for (int d = 0; d < C::D && fissued < Ftot; ++d) issue();
        advance();                                              // stage 0
    }

    size_t f = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool wave_active = ch * C::CHUNK + wave * 16 < P; // wave-uniform
        if (!wave_active) {                                     // nothing to compute: keep the barrier count (and the DMA share)
            for (size_t s = 0; s < S; ++s) {
                __builtin_amdgcn_s_barrier();
                ++f;
                if (f < Ftot) advance();
            }
            continue;
        }
        // lanes past P duplicate point P - 1 (same inputs, same outputs, same addresses): the stores of the sweep
        // need no per-lane predicate and the MFMA + epilogue body stays one basic block
        int pl = ch * C::CHUNK + wave * 16 + fr;
        pl = pl < P ? pl : P - 1;
        double xf[NK];
#pragma unroll
        for (int s = 0; s < NK; ++s) xf[s] = X[(size_t)pl * C::N + 8 * (s >> 1) + 2 * fq + (s & 1)];
        double* yp = DIFF ? Y + (pl >> 1) : (RM ? Y + pl : Y + (size_t)pl * m);
        const size_t ldr = DIFF ? (size_t)(P >> 1) : (RM ? (size_t)P : 1);   // distance between consecutive rows of one point

        // TILES == 2: acc0 / acc1 are the two row tiles of the stage; TILES == 1: the even / odd column pairs of the
        // one tile (two independent chains either way), summed into acc0 at the end
        auto mfma_stage = [&](size_t ff, Acc& acc0, Acc& acc1) {
            const unsigned char* slot = smem + (ff % C::NS) * C::STAGE_BYTES;
            acc0 = Acc{0, 0, 0, 0};
            acc1 = Acc{0, 0, 0, 0};
            if constexpr (C::TILES == 2) {
#pragma unroll
                for (int j = 0; j < NK / 2; ++j) {
                    const int off = laddr[j & 3] + (j >> 2) * 256;  // (4 j & ~15) * 16 bytes
                    const double2 a0 = *reinterpret_cast<const double2*>(slot + off);
                    const double2 a1 = *reinterpret_cast<const double2*>(slot + off + 16 * C::N * 8);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, xf[2 * j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, xf[2 * j], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, xf[2 * j + 1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, xf[2 * j + 1], acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NK / 2; j += 2) {
                    const double2 a0 = *reinterpret_cast<const double2*>(slot + laddr[j & 3] + (j >> 2) * 256);
                    const double2 a1 = *reinterpret_cast<const double2*>(slot + laddr[(j + 1) & 3] + ((j + 1) >> 2) * 256);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, xf[2 * j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, xf[2 * j + 2], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, xf[2 * j + 1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, xf[2 * j + 3], acc1, 0, 0, 0);
                }
                acc0 += acc1;
            }
        };
        // rows row0 + 2 fq + {0, 1} and row0 + 8 + 2 fq + {0, 1} of one 16-row tile
        auto epilogue_tile = [&](const Acc& acc, size_t row0, const unsigned char* bslot, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            const double2 b0 = *reinterpret_cast<const double2*>(bslot + 16 * fq);
            const double2 b1 = *reinterpret_cast<const double2*>(bslot + 64 + 16 * fq);
            const double y0 = dtanh(acc[0]) - b0.x, y1 = dtanh(acc[1]) - b0.y;
            const double y2 = dtanh(acc[2]) - b1.x, y3 = dtanh(acc[3]) - b1.y;
            const size_t ra = row0 + 2 * fq, rb = row0 + 8 + 2 * fq;
            if constexpr (DIFF) {
                const double d0 = y0 - lane_pair_swap(y0), d1 = y1 - lane_pair_swap(y1);     // f(x + h e_j) - f(x - h e_j), LS:1041 + 1045
                const double d2 = y2 - lane_pair_swap(y2), d3 = y3 - lane_pair_swap(y3);
                if ((fr & 1) == 0) {
                    if (FULL || ra < m) yp[ra * ldr] = d0;
                    if (FULL || ra + 1 < m) yp[(ra + 1) * ldr] = d1;
                    if (FULL || rb < m) yp[rb * ldr] = d2;
                    if (FULL || rb + 1 < m) yp[(rb + 1) * ldr] = d3;
                }
            } else if constexpr (RM) {
                if (FULL || ra < m) yp[ra * ldr] = y0;
                if (FULL || ra + 1 < m) yp[(ra + 1) * ldr] = y1;
                if (FULL || rb < m) yp[rb * ldr] = y2;
                if (FULL || rb + 1 < m) yp[(rb + 1) * ldr] = y3;
            } else if constexpr (FULL && ALIGNED) {
                *reinterpret_cast<double2*>(yp + ra) = make_double2(y0, y1);
                *reinterpret_cast<double2*>(yp + rb) = make_double2(y2, y3);
            } else if constexpr (FULL) {
                yp[ra] = y0;
                yp[ra + 1] = y1;
                yp[rb] = y2;
                yp[rb + 1] = y3;
            } else {
                if (ra < m) yp[ra] = y0;
                if (ra + 1 < m) yp[ra + 1] = y1;
                if (rb < m) yp[rb] = y2;
                if (rb + 1 < m) yp[rb + 1] = y3;
            }
        };
        auto epilogue = [&](const Acc& e0, const Acc& e1, size_t s, size_t ff, auto full_tag) {
            const size_t row0 = (blockIdx.x + s * (size_t)gridDim.x) * C::ROWS;
            const unsigned char* bs = smem + C::B_OFF + (ff % C::NSB) * 256;
            epilogue_tile(e0, row0, bs, full_tag);
            __builtin_amdgcn_sched_barrier(0);                  // one tile at a time: bounds the live tanh temporaries
            if constexpr (C::TILES == 2) {
                epilogue_tile(e1, row0 + 16, bs + 128, full_tag);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // The two compute waves of a SIMD (w and w + 4) run out of phase: waves 0..3 do MFMA(s) then epilogue(s),
        // waves 4..7 do epilogue(s - 1) then MFMA(s), so one wave's tanh / store VALU work overlaps the other's
        // MFMA chains without relying on instruction scheduling. Only the last stage of a workgroup can be
        // partial (stages ascend).
        Acc acc0, acc1;
        // (SELF: the wait for this wave's share of the next stage follows the matrix-core chain and precedes the stores)
        if constexpr (GROUP == 0) {
            for (size_t s = 0; s + 1 < S; ++s, ++f) {
                __builtin_amdgcn_s_barrier();                   // stage f is complete in LDS
                mfma_stage(f, acc0, acc1);
                advance();
                epilogue(acc0, acc1, s, f, std::true_type{});
            }
            __builtin_amdgcn_s_barrier();
            mfma_stage(f, acc0, acc1);
            if (f + 1 < Ftot) advance();
            epilogue(acc0, acc1, S - 1, f, std::false_type{});
            ++f;
        } else {
            __builtin_amdgcn_s_barrier();
            mfma_stage(f, acc0, acc1);
            ++f;
            if (f < Ftot) advance();
            for (size_t s = 1; s < S; ++s, ++f) {
                __builtin_amdgcn_s_barrier();
                epilogue(acc0, acc1, s - 1, f - 1, std::true_type{});
                mfma_stage(f, acc0, acc1);
                if (f + 1 < Ftot) advance();
            }
            epilogue(acc0, acc1, S - 1, f - 1, std::false_type{});
        }
    }
}

// ---- difference panel with A read ONCE (p == 2n finite-difference points X = [x + h e_0, x - h e_0, x + h e_1, ...], the
//      contract of mir_lsq_gpu_options.fbRowMajorDiff). The stage loop is the OUTER loop and the point chunks the inner one: a
//      32-row stage of A is DMA'd into LDS once and multiplied with every chunk of points before the ring moves on (the sweep
//      above streams A once per 128-point chunk: 4.1 GB instead of 3.1 at n = 128). What made that impossible for general X
//      is the B operand: 64 VGPRs of X fragments per chunk. Here every row of X is x except in ONE coordinate, so a lane keeps
//      the fragments of x itself (read from rows of X that leave the coordinate alone) plus, per chunk, the one perturbed value
//      and where it goes: the operand of k-step s is a select between the two -- the same numbers the chunked sweep feeds the
//      matrix cores, every product of the dense GEMM is still computed.
template <int NK, int GROUP>
__device__ __forceinline__ void tlb_compute_once(const double* __restrict__ X, double* __restrict__ D, size_t m,
                                                 unsigned char* smem, int lane, int wave, size_t S)
{
    using C = TlbCfg<NK>;
    using Acc = __attribute__((ext_vector_type(4))) double;
    constexpr int N = C::N, P = 2 * C::N;
    constexpr int NCH = (P + C::CHUNK - 1) / C::CHUNK;
    const int fr = lane & 15, fq = lane >> 4;
    const int v = fq ^ tlb_sigma(fr);
    int laddr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) laddr[k] = fr * N * 8 + ((4 * k) ^ v) * 16;

    // fragments of the base point: k-step s multiplies column k = 8 (s >> 1) + 2 fq + (s & 1); rows 2k and 2k + 1 of X perturb it
    double xb[NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int k = 8 * (s >> 1) + 2 * fq + (s & 1);
        xb[s] = X[(size_t)((2 * k + 2) % P) * N + k];
    }
    // per chunk: this lane's point, the k-step that carries its perturbed coordinate (if this lane's k-slot has it), the value
    bool act[NCH], hit[NCH];
    int sstar[NCH];
    double xs[NCH];
    double* dp[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        act[c] = c * C::CHUNK + wave * 16 < P;                  // wave-uniform
        int pl = c * C::CHUNK + wave * 16 + fr;
        pl = pl < P ? pl : P - 1;
        const int j = pl >> 1;
        hit[c] = fq == ((j >> 1) & 3);
        sstar[c] = 2 * (j >> 3) + (j & 1);
        xs[c] = X[(size_t)pl * N + j];
        dp[c] = D + j;
    }
    constexpr size_t ldr = N;                                   // D is m x n row-major

    auto mfma_stage = [&](size_t st, auto cc, Acc& acc0, Acc& acc1) {
        constexpr int c = decltype(cc)::value;
        const unsigned char* slot = smem + (st % C::NS) * C::STAGE_BYTES;
        acc0 = Acc{0, 0, 0, 0};
        acc1 = Acc{0, 0, 0, 0};
        auto xop = [&](int s) { return (hit[c] && sstar[c] == s) ? xs[c] : xb[s]; };
        if constexpr (C::TILES == 2) {
#pragma unroll
            for (int j = 0; j < NK / 2; ++j) {
                const int off = laddr[j & 3] + (j >> 2) * 256;
                const double2 a0 = *reinterpret_cast<const double2*>(slot + off);
                const double2 a1 = *reinterpret_cast<const double2*>(slot + off + 16 * N * 8);
                const double x0 = xop(2 * j), x1 = xop(2 * j + 1);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, x0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, x0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, x1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, x1, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NK / 2; j += 2) {
                const double2 a0 = *reinterpret_cast<const double2*>(slot + laddr[j & 3] + (j >> 2) * 256);
                const double2 a1 = *reinterpret_cast<const double2*>(slot + laddr[(j + 1) & 3] + ((j + 1) >> 2) * 256);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, xop(2 * j), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, xop(2 * j + 2), acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, xop(2 * j + 1), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, xop(2 * j + 3), acc1, 0, 0, 0);
            }
            acc0 += acc1;
        }
    };
    auto epilogue_tile = [&](const Acc& acc, size_t row0, const unsigned char* bslot, double* yp, bool full) {
        const double2 b0 = *reinterpret_cast<const double2*>(bslot + 16 * fq);
        const double2 b1 = *reinterpret_cast<const double2*>(bslot + 64 + 16 * fq);
        const double y0 = dtanh(acc[0]) - b0.x, y1 = dtanh(acc[1]) - b0.y;
        const double y2 = dtanh(acc[2]) - b1.x, y3 = dtanh(acc[3]) - b1.y;
        const double d0 = y0 - lane_pair_swap(y0), d1 = y1 - lane_pair_swap(y1);         // f(x + h e_j) - f(x - h e_j), LS:1041 + 1045
        const double d2 = y2 - lane_pair_swap(y2), d3 = y3 - lane_pair_swap(y3);
        const size_t ra = row0 + 2 * fq, rb = row0 + 8 + 2 * fq;
        if ((fr & 1) == 0) {
            if (full || ra < m) yp[ra * ldr] = d0;
            if (full || ra + 1 < m) yp[(ra + 1) * ldr] = d1;
            if (full || rb < m) yp[rb * ldr] = d2;
            if (full || rb + 1 < m) yp[(rb + 1) * ldr] = d3;
        }
    };
    auto epilogue = [&](const Acc& e0, const Acc& e1, size_t st, auto cc) {
        constexpr int c = decltype(cc)::value;
        const size_t row0 = (blockIdx.x + st * (size_t)gridDim.x) * C::ROWS;
        const unsigned char* bs = smem + C::B_OFF + (st % C::NSB) * 256;
        const bool full = st + 1 < S;                            // only the last stage of a workgroup can be partial
        epilogue_tile(e0, row0, bs, dp[c], full);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C::TILES == 2) {
            epilogue_tile(e1, row0 + 16, bs + 128, dp[c], full);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Same phase scheme as tlb_compute over the flat sequence (stage 0, chunk 0), (stage 0, chunk 1), ..., one barrier per STAGE.
    Acc acc0, acc1;
    if constexpr (GROUP == 0) {
        for (size_t st = 0; st < S; ++st) {
            __builtin_amdgcn_s_barrier();                       // stage st is complete in LDS
            wl_static_for<NCH>([&](auto cc) {
                if (act[decltype(cc)::value]) {
                    mfma_stage(st, cc, acc0, acc1);
                    epilogue(acc0, acc1, st, cc);
                }
            });
        }
    } else {
        for (size_t st = 0; st < S; ++st) {
            wl_static_for<NCH>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (c == 0) {
                    __builtin_amdgcn_s_barrier();
                    if (st > 0) {                               // the pending tile is (st - 1, last active chunk)
                        wl_static_for<NCH>([&](auto pp) {
                            constexpr int pc = decltype(pp)::value;
                            constexpr bool last = pc == NCH - 1;
                            if (act[pc] && (last || !act[pc + (last ? 0 : 1)])) epilogue(acc0, acc1, st - 1, pp);
                        });
                    }
                    if (act[0]) mfma_stage(st, cc, acc0, acc1);
                } else {
                    if (act[c]) {                               // act[c] implies act[c - 1]: a wave's chunks fill up from 0
                        epilogue(acc0, acc1, st, std::integral_constant<int, c - 1>{});
                        mfma_stage(st, cc, acc0, acc1);
                    }
                }
            });
        }
        if (S > 0) {
            wl_static_for<NCH>([&](auto pp) {
                constexpr int pc = decltype(pp)::value;
                constexpr bool last = pc == NCH - 1;
                if (act[pc] && (last || !act[pc + (last ? 0 : 1)])) epilogue(acc0, acc1, S - 1, pp);
            });
        }
    }
}

template <int NK, bool RM = false, bool DIFF = false>
__global__ __launch_bounds__(TlbCfg<NK>::THREADS) void k_tanh_linear_batched_dma(const double* __restrict__ A,
                                                                                  const double* __restrict__ b,
                                                                                  const double* __restrict__ X,
                                                                                  double* __restrict__ Y, size_t m, int P,
                                                                                  int read_a_once)
{
    using C = TlbCfg<NK>;
    extern __shared__ __attribute__((aligned(16))) unsigned char tlb_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t Stot = (m + C::ROWS - 1) / C::ROWS;
    const size_t S = blockIdx.x < Stot ? (Stot - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;   // stages blockIdx.x + k grid
    const int nchunks = (P + C::CHUNK - 1) / C::CHUNK;
    // the finite-difference points of fbRowMajorDiff, on request: A is read once (2.1 instead of 3.1 GB per call at n = 128,
    // and 3 % SLOWER: the kernel is MFMA-bound and the operand selects are extra VALU work -- not the default)
    // (n = 256: six chunks of per-chunk state next to 128 VGPRs of fragments spill -- measured 2x slower; not offered there)
    const bool once = DIFF && read_a_once && NK <= 32 && P == 2 * C::N;
    const size_t F = once ? S : S * (size_t)nchunks;            // flat (chunk, stage) sequence: the ring never drains
    if (S == 0) return;
    if (!C::SELF && wave == C::COMPUTE_WAVES) tlb_loader<NK, 0>(A, b, m, tlb_smem, lane, S, F);
    else if (!C::SELF && wave == C::COMPUTE_WAVES + 1) tlb_loader<NK, 1>(A, b, m, tlb_smem, lane, S, F);
    else if (DIFF && once) {
        if constexpr (DIFF) {
            if (wave < 4) tlb_compute_once<NK, 0>(X, Y, m, tlb_smem, lane, wave, S);     // waves w and w + 4 share a SIMD
            else tlb_compute_once<NK, 1>(X, Y, m, tlb_smem, lane, wave, S);
        }
    } else {
        if constexpr (RM) {
            if (wave < 4) tlb_compute<NK, false, 0, true, DIFF>(A, b, X, Y, m, P, tlb_smem, lane, wave, S, nchunks);
            else tlb_compute<NK, false, 1, true, DIFF>(A, b, X, Y, m, P, tlb_smem, lane, wave, S, nchunks);
        } else {
            const bool aligned = ((m & 1) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
            if (wave < 4) {
                if (aligned) tlb_compute<NK, true, 0>(A, b, X, Y, m, P, tlb_smem, lane, wave, S, nchunks);
                else tlb_compute<NK, false, 0>(A, b, X, Y, m, P, tlb_smem, lane, wave, S, nchunks);
            } else {
                if (aligned) tlb_compute<NK, true, 1>(A, b, X, Y, m, P, tlb_smem, lane, wave, S, nchunks);
                else tlb_compute<NK, false, 1>(A, b, X, Y, m, P, tlb_smem, lane, wave, S, nchunks);
            }
        }
    }
}

template <int NK, bool RM = false, bool DIFF = false>
bool launch_tlb_dma(const double* A, const double* b, const double* X, double* Y, size_t m, int P, hipStream_t s, int read_a_once = 0)
{
    using C = TlbCfg<NK>;
    static bool attr_ok = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k_tanh_linear_batched_dma<NK, RM, DIFF>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) == hipSuccess;
    }();
    if (!attr_ok) return false;
    const size_t Stot = (m + C::ROWS - 1) / C::ROWS;
    const unsigned grid = (unsigned)(Stot < 256 ? Stot : 256);
    hipLaunchKernelGGL((k_tanh_linear_batched_dma<NK, RM, DIFF>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, s, A, b, X, Y, m, P, read_a_once);
    return true;
}

// row-major output for shapes the DMA kernel does not cover: one thread per (row, point), plain dot product
__global__ __launch_bounds__(256) void k_tanh_linear_batched_rm_generic(const double* __restrict__ A, const double* __restrict__ b,
                                                                        const double* __restrict__ X, double* __restrict__ Y,
                                                                        size_t m, int n, int P)
{
    const size_t total = m * (size_t)P;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / P;
        const int k = (int)(e % P);
        const double* a = A + i * (size_t)n;
        const double* x = X + (size_t)k * n;
        double s0 = 0;
        for (int j = 0; j < n; ++j) s0 += a[j] * x[j];
        Y[e] = dtanh(s0) - b[i];
    }
}

// difference panel for shapes the DMA kernel does not cover: one thread per (row, column), two dot products
__global__ __launch_bounds__(256) void k_tanh_linear_batched_diff_generic(const double* __restrict__ A, const double* __restrict__ b,
                                                                          const double* __restrict__ X, double* __restrict__ D,
                                                                          size_t m, int n, int P)
{
    const int nc = P / 2;
    const size_t total = m * (size_t)nc;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / nc;
        const int j = (int)(e % nc);
        const double* a = A + i * (size_t)n;
        const double* xp = X + (size_t)(2 * j) * n;
        const double* xm = xp + n;
        double sp = 0, sm = 0;
        for (int k = 0; k < n; ++k) { sp += a[k] * xp[k]; sm += a[k] * xm[k]; }
        const double yp = dtanh(sp) - b[i], ym = dtanh(sm) - b[i];
        D[e] = yp - ym;
    }
}

// ---- any n above the LDS-DMA shapes (n > 256, n % 8 == 0): the same GEMM on the matrix cores with both operands read straight
//      from memory (A streamed once per workgroup: its four waves share a 16-row tile through L1 / L2; X, a few MB, lives in L2).
//      A wave owns 16 rows x 64 points (four accumulators reuse each A fragment); the K index is permuted as in the LDS-DMA kernel
//      so that one 16-byte load feeds two k-steps: lane (fr, fq) reads columns 8 j + 2 fq, + 1 of its row / of its point.
//      (Before: one thread per (row, point) with a scalar dot product, 236 ms for the 1024 points of an n = 512 refresh at
//      m = 250 000 -- 1.1 TFLOP/s; the solver's own work at that shape is 20 ms.)
// OUT: 0 = Y m x P row-major, 1 = the m x P/2 row-major DIFFERENCE panel, 2 = Y point-major (Y[k m + i])
template <int OUT>
__global__ __launch_bounds__(256) void k_tanh_linear_batched_wide(const double* __restrict__ A, const double* __restrict__ b,
                                                                    const double* __restrict__ X, double* __restrict__ Y,
                                                                    size_t m, int n, int P)
{
    // A wave owns RT = 4 row tiles x 4 point tiles (64 rows x 64 points, 128 accumulator registers): per pair of k-steps 4 + 4
    // sixteen-byte loads feed 32 MFMAs. With one row tile (1 + 4 loads for 8 MFMAs) the kernel ran at the CU's vector-memory address
    // rate and re-read its points' 256 KB of X from L2 for every 16 rows: the 1024 points of an n = 512 refresh at m = 250 000 took
    // 8.8 ms (30 TFLOP/s); two row tiles 7.4, four 6.7 (39 TFLOP/s, one wave a SIMD). The next step would be X and A through LDS.
    constexpr int RT = 4;
    using Acc = __attribute__((ext_vector_type(4))) double;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const size_t ntiles = (m + 16 * RT - 1) / (16 * RT);
    const int ngroups = (P + 63) / 64;
    const int npairs = n / 8;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const double2* __restrict__ ap[RT];
        double bv[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const size_t r0 = (t * RT + rt) * 16;
            const size_t arow = r0 + fr < m ? r0 + fr : m - 1;
            ap[rt] = reinterpret_cast<const double2*>(A + arow * (size_t)n) + fq;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const size_t row = r0 + fq + 4 * r; bv[rt][r] = b[row < m ? row : m - 1]; }
        }
        for (int g = wave; g < ngroups; g += 4) {
            const double2* __restrict__ xp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int pt = 64 * g + 16 * u + fr;
                pt = pt < P ? pt : P - 1;
                xp[u] = reinterpret_cast<const double2*>(X + (size_t)pt * n) + fq;
            }
            Acc acc[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[rt][u] = Acc{0, 0, 0, 0};
#pragma unroll 2
            for (int j = 0; j < npairs; ++j) {
                double2 a[RT], x[4];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[rt] = ap[rt][4 * j];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = xp[u][4 * j];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[rt][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt].x, x[u].x, acc[rt][u], 0, 0, 0);
                        acc[rt][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt].y, x[u].y, acc[rt][u], 0, 0, 0);
                    }
            }
            // D: column = lane & 15 = point, rows fq + 4 r
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pt = 64 * g + 16 * u + fr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t row = (t * RT + rt) * 16 + fq + 4 * r;
                        const double y = dtanh(acc[rt][u][r]) - bv[rt][r];
                        if constexpr (OUT == 1) {
                            const double d = y - lane_pair_swap(y);              // f(x + h e_j) - f(x - h e_j): points 2 j, 2 j + 1 are adjacent lanes
                            if ((fr & 1) == 0 && pt < P && row < m) Y[row * (size_t)(P >> 1) + (pt >> 1)] = d;
                        } else if constexpr (OUT == 0) {
                            if (pt < P && row < m) Y[row * (size_t)P + pt] = y;
                        } else {
                            if (pt < P && row < m) Y[(size_t)pt * m + row] = y;
                        }
                    }
                }
        }
    }
}

}  // namespace

void launch_tanh_linear_batched_diff(const double* A, const double* b, const double* X, double* D, size_t m, int n, int P,
                                     hipStream_t s, int read_a_once)
{
    if (m >= 32 && P % 16 == 0) {                          // whole 16-point MFMA tiles: no clamped lanes, whose pairs would store zeros
        if (n == 256 && launch_tlb_dma<64, true, true>(A, b, X, D, m, P, s, read_a_once)) return;
        if (n == 128 && launch_tlb_dma<32, true, true>(A, b, X, D, m, P, s, read_a_once)) return;
        if (n == 64 && launch_tlb_dma<16, true, true>(A, b, X, D, m, P, s, read_a_once)) return;
        if (n == 32 && launch_tlb_dma<8, true, true>(A, b, X, D, m, P, s, read_a_once)) return;
    }
    if (n > 256 && n % 8 == 0 && P % 2 == 0 && m > 0) {
        const size_t nt = (m + 63) / 64;
        hipLaunchKernelGGL(k_tanh_linear_batched_wide<1>, dim3((unsigned)(nt < 256 * 8 ? nt : 256 * 8)), dim3(256), 0, s, A, b, X, D, m, n, P);
        return;
    }
    size_t blocks = (m * (size_t)(P / 2) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_tanh_linear_batched_diff_generic, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, s, A, b, X, D, m, n, P);
}

void launch_tanh_linear_batched_rm(const double* A, const double* b, const double* X, double* Y, size_t m, int n, int P,
                                   hipStream_t s)
{
    if (m >= 32) {
        if (n == 256 && launch_tlb_dma<64, true>(A, b, X, Y, m, P, s)) return;
        if (n == 128 && launch_tlb_dma<32, true>(A, b, X, Y, m, P, s)) return;
        if (n == 64 && launch_tlb_dma<16, true>(A, b, X, Y, m, P, s)) return;
        if (n == 32 && launch_tlb_dma<8, true>(A, b, X, Y, m, P, s)) return;
    }
    if (n > 256 && n % 8 == 0 && m > 0) {
        const size_t nt = (m + 63) / 64;
        hipLaunchKernelGGL(k_tanh_linear_batched_wide<0>, dim3((unsigned)(nt < 256 * 8 ? nt : 256 * 8)), dim3(256), 0, s, A, b, X, Y, m, n, P);
        return;
    }
    size_t blocks = (m * (size_t)P + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_tanh_linear_batched_rm_generic, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, s, A, b, X, Y, m, n, P);
}

bool launch_tanh_linear_batched(const double* A, const double* b, const double* X, double* Y, size_t m, int n, int P,
                                hipStream_t s)
{
    if (n == 256 && m >= 32 && launch_tlb_dma<64>(A, b, X, Y, m, P, s)) return true;
    if (n > 256 && n % 8 == 0 && m > 0) {
        const size_t nt = (m + 63) / 64;
        hipLaunchKernelGGL(k_tanh_linear_batched_wide<2>, dim3((unsigned)(nt < 256 * 8 ? nt : 256 * 8)), dim3(256), 0, s, A, b, X, Y, m, n, P);
        return true;
    }
    if (n % 2 != 0 || n > 128 || n < 4) return false;
    if (m >= 32) {
        if (n == 128 && launch_tlb_dma<32>(A, b, X, Y, m, P, s)) return true;
        if (n == 64 && launch_tlb_dma<16>(A, b, X, Y, m, P, s)) return true;
        if (n == 32 && launch_tlb_dma<8>(A, b, X, Y, m, P, s)) return true;
    }
    const size_t ntiles = (m + 15) / 16;
    unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    if (n <= 16) hipLaunchKernelGGL(k_tanh_linear_batched<4>, dim3(grid), dim3(1024), 0, s, A, b, X, Y, m, n, P);
    else if (n <= 32) hipLaunchKernelGGL(k_tanh_linear_batched<8>, dim3(grid), dim3(1024), 0, s, A, b, X, Y, m, n, P);
    else if (n <= 64) hipLaunchKernelGGL(k_tanh_linear_batched<16>, dim3(grid), dim3(1024), 0, s, A, b, X, Y, m, n, P);
    else hipLaunchKernelGGL(k_tanh_linear_batched<32>, dim3(grid), dim3(1024), 0, s, A, b, X, Y, m, n, P);
    return true;
}
