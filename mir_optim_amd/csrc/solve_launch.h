// solve_launch.h -- host view of the n x n kernels (solve_kernel.h / solve_lds.h for n <= 256, solve_big.h for any n):
// launch entry points, defined in launch_solve_d.hip / launch_solve_s.hip (one translation unit per element type: these
// kernels are the largest in the library). Argument blocks: solve_types.h.
// Reference operations replaced: least_squares.d:1053-1110, 1141-1142, 1164 and boxcqp.d:122-379 (solveBoxQP with
// mir-lapack's posvx('E','L')).
#pragma once

#include "common.h"
#include "solve_types.h"

namespace mirlsq {

// one workgroup per ladder entry (ks <= kChainMax). bounded = false: every lower / upper entry is infinite (the BOXCQP
// active-set loop is compiled out); generic = true: the any-n kernel of solve_big.h (required above n = kSolveMaxN)
template <typename T>
hipError_t launch_lm_solve(const LmSolveArgs<T>& a, int ks, bool bounded, bool generic, hipStream_t s);

// standalone BOXCQP on device-resident operands (mir_solve_box_qp_gpu_*): picks the kernel by a.n
template <typename T>
hipError_t launch_box_qp(const BoxQpArgs<T>& a, hipStream_t s);

}  // namespace mirlsq
