import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
from oracle import oracle as O
import problems as P
import test_gpu_lm as T
for (m, n, mode) in [(4096, 16, "fd"), (20000, 32, "batched"), (50000, 128, "batched")]:
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    so = O.default_settings(); so.absTolerance = 1e-9
    tr = M.Trace()
    res, x = prob.solve(w["x0"].copy(), settings=s, trace=tr, batched=(mode == "batched"))
    ro, xo, ev = T._oracle_trace(O, w, so)
    got = tr.records()
    print(m, n, mode, res, O.STATUS[ro.status], ro.iterations, ro.fCalls, len(got), len(ev))
    for k in range(max(len(got), len(ev))):
        g = got[k] if k < len(got) else None
        e = ev[k] if k < len(ev) else None
        fmt = lambda r: "%d it%2d lam %.6e res %.12e tr %.12e dx %.3e" % r if r else "-"
        print("%3d  G %s\n     O %s" % (k, fmt(g), fmt(e)))
