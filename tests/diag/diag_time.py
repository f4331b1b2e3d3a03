"""Diagnostic: time GPU solve vs oracle solve on tanh-linear problems (run on the GPU box)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
from oracle import oracle as O
import problems as P

for m, n in [(512, 8), (4096, 16), (50000, 128)]:
    w = P.tanh_linear(m, n)
    t0 = time.time(); prob = W.TanhLinear(w["A"], w["b"]); t1 = time.time()
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    for rep in range(2):
        st = M.Stats()
        t2 = time.time(); res, x = prob.solve(w["x0"], settings=s, stats=st, flags=M.TIME_KERNELS); t3 = time.time()
        print(m, n, "gpu solve %.3fs" % (t3 - t2), res, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.as_dict().items()})
    so = O.default_settings(); so.absTolerance = 1e-9
    ctx = O.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    t4 = time.time(); ro, xo = O.optimize(O.native_fn("wlc_tanh_linear_f"), m, w["x0"], settings=so, fctx=C.addressof(ctx)); t5 = time.time()
    print(m, n, "setup %.3fs oracle %.3fs" % (t1 - t0, t5 - t4), O.STATUS[ro.status], ro.iterations, ro.fCalls, "maxdiff", np.abs(x - xo).max())
