// jtj_plan.h -- host view of the J^T J kernels: their argument block, which kernel runs for a shape (JtjPlan) and the
// launch entry points (defined in launch_jtj.hip, the only translation unit that instantiates the kernels).
//
// Kernel by shape, LS = /root/reference/source/mir/optim/least_squares.d:
//   f64, n <= 128               k_jtj_fdp<NCB, true>    finite-difference panel -> J, J^T J, J^T y   (LS:1041-1047, 1052, 1065)
//   f64, n <= 128               k_jtj_fdp<NCB, false>   J^T J + J^T y of a given J (odd n: 8-byte loads) (LS:1052, 1065)
//   f64, 128 < n <= 256 (any n, m)             k_jtj_fdp8   finite-difference panel -> J, J^T J, J^T y
//   f64, 128 < n <= 256 (n % 16 == 0, m even)  k_jtj8   eight-wave LDS-DMA ring, J^T J of a given J
//   f64, 128 < n <= 256, the other n and m     k_jtj_fdp8<., false, true>   J^T J of a given J (flat buffer loads)
//   f32, n <= 128, n % 4 == 0   k_jtj_pc32
//   everything else             k_jtj (n <= 128, register streaming) / k_jtj_wide (any n: 64-column tile pairs)
// MIR_LSQ_VARIANT_BROYDEN_REWRITE runs the Broyden pass as the literal restatement of LS:1003-1006 (J rewritten):
// k_jtj<., ., true>, k_jtj8 with its Broyden flag, k_broyden_wide / k_broyden_rows in front of the tile-pair jobs.
#pragma once

#include "../../include/mir_optim_amd.h"
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

// tile-pair jobs (jtj_wide.h)
constexpr int kWideTile = 4;                                             // blocks per tile side
constexpr int kWideSlabLen = (kWideTile * kWideTile * 4 + kWideTile) * kWave;   // 16 blocks x 4 regs + 4 jy regs, x 64 lanes
// dynamic LDS of k_jtj8<NCB> (jtj_ring8.h: four 16-row stages + the y / y_old rings; launch_jtj.hip asserts the match)
constexpr size_t jtj8_lds_bytes(int ncb) { return (size_t)4 * 16 * 16 * ncb * 8 + 2 * 4 * 1024; }

// Where the slab reduction may put its result besides `packed`: the full symmetric J^T J and J^T y (the work of
// k_unpack_grad; max |J^T y| is taken by the solve kernel), when no all-reduce of `packed` sits in between. All null: `packed` only.
template <typename T>
struct JtjUnpack {
    T* JJ = nullptr; T* Jy = nullptr;
};

struct JtjPlan {
    int ncb = 0;
    int nblk = 0;
    int slab_len = 0;
    size_t lds = 0;
    bool wide = false;      // n > 128 and not ring8: 64-column tile-pair jobs (jtj_wide.h), any n
    bool ring8 = false;     // 128 < n <= 256, f64, n % 16 == 0, m even: eight-wave LDS-DMA ring (jtj_ring8.h)
    bool fdp = false;       // f64, n <= 128, any m: producer / consumer kernel (jtj_fdp.h) for the finite-difference J^T J
    bool fdp_plain = false; // ... and for the plain J^T J / the difference panel (odd n: the element-wise producer)
    bool fdp8 = false;      // f64, 128 < n <= 256, any n and m: eight producer + consumer waves (jtj_fdp8.h), FD J^T J only
    int fdp8_nblk = 0, fdp8_slab_len = 0, fdp8_ncb = 0;     // fdp8_ncb: the even block count the kernel is compiled for (>= ncb)
    bool pc32 = false;      // f32, n <= 128, n % 4 == 0, any m: producer / consumer kernel on v_mfma_f32_16x16x4 (jtj_pc32.h), plain J^T J
    int pc32_nblk = 0;
    int njobs = 1;
    int stream_nblk = 0;    // n <= 128: grid and dynamic LDS of the register-streaming kernel k_jtj
    size_t stream_lds = 0;
};

template <typename T>
JtjPlan jtj_plan(size_t m, int n, int num_cu)
{
    JtjPlan p;
    p.ncb = (n + 15) / 16;
    const int nacc = p.ncb * (p.ncb + 1) / 2;
    p.slab_len = (nacc * 4 + p.ncb) * kWave;
    if (sizeof(T) == 8 && n > 128 && n <= 256) {
        // any n: the kernel is compiled for n rounded up to a multiple of 32 (an even number of 16-column blocks)
        p.fdp8 = true;
        p.fdp8_ncb = 2 * ((n + 31) / 32);
        const size_t stot = (m + 15) / 16;
        const size_t want = (stot + 7) / 8;                    // at least ~8 stages per workgroup
        p.fdp8_nblk = (int)(want < (size_t)num_cu ? (want ? want : 1) : (size_t)num_cu);   // one workgroup per CU
        p.fdp8_slab_len = (p.fdp8_ncb * (p.fdp8_ncb + 1) / 2 * 4 + p.fdp8_ncb) * kWave;
    }
    if (sizeof(T) == 8 && n > 128 && n <= 256 && n % 16 == 0 && m % 2 == 0) {
        p.ring8 = true;
        p.lds = jtj8_lds_bytes(p.ncb);
        const size_t stot = (m + 15) / 16;
        size_t want = (stot + 7) / 8;                          // at least ~8 stages per workgroup
        p.nblk = (int)(want < (size_t)num_cu ? (want ? want : 1) : (size_t)num_cu);   // one workgroup per CU
        return p;
    }
    if (n > 128) {
        p.wide = true;
        const int nt = (p.ncb + kWideTile - 1) / kWideTile;
        p.njobs = nt * (nt + 1) / 2;
        p.slab_len = kWideSlabLen;
        p.lds = (size_t)2 * kWideSlabLen * sizeof(T);
        const size_t G = (m + 3) / 4;
        size_t want = (G + 4 * 8 - 1) / (4 * 8);
        size_t cap = (size_t)num_cu * 4 / p.njobs;
        if (cap < 1) cap = 1;
        p.nblk = (int)(want < cap ? (want ? want : 1) : cap);
        return p;
    }
    if (sizeof(T) == 4 && n <= 128 && n % 4 == 0) {
        p.pc32 = true;
        const size_t stot = (m + 63) / 64;                     // 64-row stages
        const size_t want = (stot + 3) / 4;                    // at least ~4 stages per workgroup
        const size_t cap = (size_t)num_cu * 2;
        p.pc32_nblk = (int)(want < cap ? (want ? want : 1) : cap);
    }
    p.fdp = sizeof(T) == 8 && n <= 128;
    p.fdp_plain = p.fdp;                // any n: odd n (rows not on 16-byte boundaries) takes the element-wise producer
    // the register-streaming kernel k_jtj: f64 with odd n, f32 with n % 4 != 0, and every Broyden REWRITE at n <= 128
    {
        const int rpb = 4 * (int)(sizeof(T) / 4);
        const int roles = jtj_roles_rt(p.ncb, rpb);
        p.stream_lds = (size_t)(roles == 4 ? 0 : (roles == 2 ? 1 : 2)) * p.slab_len * sizeof(T);
        // workgroups per CU: LDS- and register-limited (one workgroup = one wave per SIMD)
        int per_cu = p.stream_lds ? (int)((160 * 1024) / p.stream_lds) : 8;
        const int reg_waves = (nacc * rpb / roles > 40) ? 2 : 4;   // matches jtj_min_waves
        if (per_cu > reg_waves) per_cu = reg_waves;
        if (per_cu < 1) per_cu = 1;
        const size_t G = (m + 3) / 4;
        const size_t slots_per_blk = kJtjWaves / roles;
        size_t want = (G + slots_per_blk * 8 - 1) / (slots_per_blk * 8);     // at least ~8 row groups per wave
        size_t cap = (size_t)num_cu * per_cu;
        p.stream_nblk = (int)(want < cap ? (want ? want : 1) : cap);
    }
    p.nblk = p.stream_nblk;         // k_jtj_fdp runs on the same grid, except:
    if (sizeof(T) == 8 && n % 16 == 0 && n <= 128 && m % 2 == 0) {
        // the shapes k_jtj_fdp was tuned on: two workgroups per CU, at least ~8 blocks of rs rows each (the partition of the
        // retired LDS-DMA ring kernel; kept: it fixes the summation order of the slabs)
        const int rs = p.ncb <= 4 ? 32 : (p.ncb == 5 ? 24 : (p.ncb == 6 ? 20 : 16));
        const size_t stot = (m + rs - 1) / rs;
        size_t want = (stot + 7) / 8;
        const size_t cap = (size_t)num_cu * 2;
        p.nblk = (int)(want < cap ? (want ? want : 1) : cap);
    }
    return p;
}

// slab elements the J^T J kernels of a plan may write
inline size_t jtj_slab_elems(const JtjPlan& a)
{
    size_t e = (size_t)a.nblk * a.njobs * a.slab_len;
    const size_t ec = (size_t)a.fdp8_nblk * a.fdp8_slab_len, ed = (size_t)a.pc32_nblk * a.slab_len;
    const size_t ef = (size_t)a.stream_nblk * a.slab_len;
    e = e > ec ? e : ec;
    e = e > ed ? e : ed;
    return e > ef ? e : ef;
}

// does jtj_run honour a JtjUnpack for this plan? (the tile-pair jobs have a reduction of their own; jtj_run_fd* always do)
inline bool jtj_plain_unpacks(const JtjPlan& p, bool broyden = false) { return !p.wide || (p.fdp8 && !broyden); }
// can the m x n DIFFERENCE panel be consumed for this shape? (f64; n <= 128: fdp_plain; n = 192, 256: k_jtj_fdp8)
inline bool jtj_fd_diff_ok(const JtjPlan& p, int n) { return p.fdp_plain || (p.fdp8 && n % 64 == 0); }

// ---- launch entry points (launch_jtj.hip; instantiated for double and float). Every kernel they launch is counted in
//      tl_launches. `packed` receives [J^T J lower | J^T y]; `u` additionally the unpacked form (see JtjUnpack).
// [Broyden rewrite, MIR_LSQ_VARIANT_BROYDEN_REWRITE] + J^T J + J^T y of the J in a.J
template <typename T>
hipError_t jtj_run(const JtjPlan& p, const JtjArgs<T>& a, bool broyden, T* packed, hipStream_t s, const JtjUnpack<T>& u = {});
// finite-difference pair panel (a.J: m x 2n row-major, a.twh) -> a.Jout, packed
template <typename T>
hipError_t jtj_run_fd(const JtjPlan& p, const JtjArgs<T>& a, T* packed, hipStream_t s, const JtjUnpack<T>& u = {});
// finite-difference DIFFERENCE panel (a.J: m x n row-major, D_ij = f(x + h e_j)_i - f(x - h e_j)_i; a.twh) -> a.Jout, packed
template <typename T>
hipError_t jtj_run_fd_diff(const JtjPlan& p, const JtjArgs<T>& a, T* packed, hipStream_t s, const JtjUnpack<T>& u = {});
// packed [J^T J lower | J^T y] -> full symmetric JJ, Jy, st->jy_inf (behind a communicator's all-reduce)
template <typename T> struct LmState;
template <typename T>
hipError_t jtj_unpack(const T* packed, int n, T* JJ, T* Jy, LmState<T>* st, hipStream_t s);

}  // namespace mirlsq
