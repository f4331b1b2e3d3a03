"""cfg 2 (Gaussian-sum fit, m = 1e5, n = 16, bounded, FD Jacobian) timing: GPU solve vs oracle (run on the GPU box)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
from oracle import oracle as O
import problems as P

g = P.gauss_sum(100000, K=5)
prob = W.Curve("gauss_sum", g["t"], g["data"])
for rep in range(3):
    st = M.Stats()
    t0 = time.perf_counter()
    res, x = prob.solve(g["x0"], g["lower"], g["upper"], stats=st, flags=M.TIME_KERNELS)
    dt = time.perf_counter() - t0
    print("gpu %.2f ms" % (dt * 1e3), res, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.as_dict().items()})
ctx = O.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
t0 = time.perf_counter()
ro, xo = O.optimize(O.native_fn("wlc_gauss_sum_f"), g["m"], g["x0"], lower=g["lower"], upper=g["upper"], fctx=C.addressof(ctx))
print("oracle %.2f ms" % ((time.perf_counter() - t0) * 1e3), O.STATUS[ro.status], ro.iterations, ro.fCalls)
