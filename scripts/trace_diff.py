"""How closely do the GPU trace and the oracle trace agree? Prints, per problem, the largest relative difference of
lambda / residual / trial residual / dx.dx over the passes before the first noise-decided one (run on the GPU box)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
from oracle import oracle as O
import problems as P
from test_gpu_lm import first_noisy_pass


def rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-300)


def report(name, got, ev):
    K = min(first_noisy_pass(got), first_noisy_pass(ev), len(got), len(ev))
    worst = [0.0] * 4
    for g, e in zip(got[:K], ev[:K]):
        if g[:2] != e[:2]:
            print(f"{name}: events differ at {g} vs {e}"); break
        for k in range(4):
            worst[k] = max(worst[k], rel(g[2 + k], e[2 + k]))
    print(f"{name:28s} passes compared {K:3d} of {len(got)}/{len(ev)}   max rel diff  lambda {worst[0]:.1e}  residual {worst[1]:.1e}  trial {worst[2]:.1e}  dx.dx {worst[3]:.1e}")


for name in ("t1", "t2", "t3a", "t3b", "t4", "t6"):
    p = getattr(P, name)()
    tr = M.Trace(); opt = M.GpuOptions(); opt.trace = C.pointer(tr.header)
    M.optimizeLeastSquares(p["f"], p["m"], np.array(p["x0"], dtype=float), p["lower"], p["upper"], g=p["g"], options=opt)
    ev = []
    O.optimize(p["f"], p["m"], p["x0"], lower=p["lower"], upper=p["upper"], g=p["g"], trace=lambda *a: ev.append(a))
    report("host-callback " + name, tr.records(), ev)
for m, n in ((5000, 8), (20000, 32), (50000, 128), (30000, 64)):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    tr = M.Trace(4096)
    prob.solve(w["x0"], settings=s, trace=tr, batched=True)
    so = O.default_settings(); so.absTolerance = 1e-9
    ctx = O.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ev = []
    O.optimize(O.native_fn("wlc_tanh_linear_f"), m, w["x0"], settings=so, fctx=C.addressof(ctx), trace=lambda *a: ev.append(a))
    report(f"device tanh-linear {m}x{n}", tr.records(), ev)
