import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mir_optim_amd as M
from oracle import oracle as O
rng = np.random.default_rng(2)
t = np.linspace(0, 1, 40); data = 2.0 * np.exp(-1.5 * t) + 0.3 + 0.01 * rng.standard_normal(40)
def f(p, y):
    y[:] = p[0] * np.exp(-p[1] * t) + p[2] - data
l, u = [-np.inf, 1.5, -np.inf], [np.inf, 1.5, np.inf]
for variant in (0, M.VARIANT_DEBUG_SOLVE):
    tr = M.Trace(1024)
    o = M.api.GpuOptions(); o.trace = __import__("ctypes").pointer(tr.header); o.variant = variant
    res, x = M.optimizeLeastSquares(f, 40, np.array([1.0, 1.5, 0.0]), l, u, settings=M.LeastSquaresSettings(), options=o)
    print("variant", variant, res, x)
    got = tr.records()
    if variant == 0:
        ev = []
        ro, xo = O.optimize(f, 40, np.array([1.0, 1.5, 0.0]), lower=l, upper=u, trace=lambda *a: ev.append(a))
        print("oracle", ro.status, ro.iterations, ro.fCalls, ro.residual, xo)
    for k in range(min(len(got), len(ev))):
        g, e = got[k], ev[k]
        same = (int(g[0]), int(g[1])) == (int(e[0]), int(e[1])) and np.isclose(g[2], e[2], rtol=1e-6) and np.allclose(g[3:5], e[3:5], rtol=1e-9, atol=1e-300)
        if not same:
            print("  first difference at", k, "\n   gpu   ", g, "\n   oracle", e)
            for j in range(max(0, k - 2), min(k + 3, len(got), len(ev))):
                print("     ", j, got[j], ev[j])
            break
    else:
        print("  traces agree on", min(len(got), len(ev)), "events; lengths", len(got), len(ev))
