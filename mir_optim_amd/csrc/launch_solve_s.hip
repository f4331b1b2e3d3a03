// launch_solve_s.hip -- the fp32 n x n kernels (see launch_solve.inc, solve_launch.h)
#define MIRLSQ_SOLVE_T float
#define MIRLSQ_SOLVE_TAG solve_s
#include "launch_solve.inc"
