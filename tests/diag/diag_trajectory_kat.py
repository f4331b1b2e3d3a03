"""Diagnostic: GPU (host-callback entry) vs oracle traces on the reference unittest problems."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from oracle import oracle as O
import problems as P
name = sys.argv[1] if len(sys.argv) > 1 else "t2"
p = getattr(P, name)()
tr = M.Trace()
opt = M.GpuOptions()
opt.trace = C.pointer(tr.header)
res, x = M.optimizeLeastSquares(p["f"], p["m"], np.array(p["x0"], dtype=float), p["lower"], p["upper"], g=p["g"], options=opt)
ev = []
ro, xo = O.optimize(p["f"], p["m"], p["x0"], lower=p["lower"], upper=p["upper"], g=p["g"], trace=lambda *a: ev.append(a))
got = tr.records()
print(res, O.STATUS[ro.status], ro.iterations, ro.fCalls, len(got), len(ev))
for k in range(max(len(got), len(ev))):
    g = got[k] if k < len(got) else None
    e = ev[k] if k < len(ev) else None
    fmt = lambda r: "%d it%2d lam %.16e res %.16e tr %.16e dx %.16e" % r if r else "-"
    print("%3d  G %s\n     O %s" % (k, fmt(g), fmt(e)))
