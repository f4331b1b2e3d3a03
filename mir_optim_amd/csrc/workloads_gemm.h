// workloads_gemm.h -- launchers of the batched tanh-linear residual GEMM (workloads_gemm.hip) for the C entries in workloads.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

// Y[p][i] = tanh(a_i . X[p]) - b_i, point-major; false when the shape is not covered (caller loops over the points)
bool launch_tanh_linear_batched(const double* A, const double* b, const double* X, double* Y, size_t m, int n, int P, hipStream_t s);
// the same with Y written m x P row-major
void launch_tanh_linear_batched_rm(const double* A, const double* b, const double* X, double* Y, size_t m, int n, int P, hipStream_t s);
// P = 2n finite-difference points: the m x n row-major DIFFERENCE panel D[i n + j] = f(X_2j)_i - f(X_2j+1)_i
void launch_tanh_linear_batched_diff(const double* A, const double* b, const double* X, double* D, size_t m, int n, int P, hipStream_t s,
                                     int read_a_once);
