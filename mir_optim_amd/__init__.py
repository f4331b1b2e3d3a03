"""mir_optim_amd -- MI355X-native Levenberg-Marquardt least squares, drop-in for the
`mir.optim.least_squares` path of libmir/mir-optim (see include/mir_optim_amd.h).

The Python layer is a thin ctypes binding over the C ABI that mirrors the reference's D API
names (LeastSquaresSettings, LeastSquaresResult, optimize, optimizeLeastSquares, solveBoxQP ...).
It never computes: if the HIP library is missing, importing `mir_optim_amd.api` raises.
"""
from .api import *  # noqa: F401,F403
