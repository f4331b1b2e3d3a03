// launch_solve_d.hip -- the fp64 n x n kernels (see launch_solve.inc, solve_launch.h)
#define MIRLSQ_SOLVE_T double
#define MIRLSQ_SOLVE_TAG solve_d
#include "launch_solve.inc"
