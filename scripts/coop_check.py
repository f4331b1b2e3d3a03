"""The any-n solve with and without its helper workgroups (MIR_LSQ_VARIANT_SOLVE_ONE_WORKGROUP) on small tanh-linear problems
above n = 256: results side by side, solve-kernel time per launch. usage: python scripts/coop_check.py [n ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P

for n in [int(a) for a in sys.argv[1:]] or [300, 512, 1024]:
    w = P.tanh_linear(max(5000, 8 * n), n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-7
    lo = np.full(n, -np.inf); up = np.full(n, np.inf)
    up[::7] = w["x0"][::7] + 0.01                      # a few bounds, some of them binding
    out = []
    for variant in (M.VARIANT_SOLVE_ONE_WORKGROUP, 0):
        st = M.Stats()
        res, x = prob.solve(np.minimum(w["x0"], up), lo, up, settings=s, batched=True, stats=st, flags=M.TIME_KERNELS, variant=variant)
        out.append((res, x))
        print(f"n={n} variant={variant}: {res}  solve kernel {st.solve_ms / max(1, st.solve_launches) * 1e3:.1f} us x {st.solve_launches}", flush=True)
    print(f"   |x_coop - x_one|max = {np.abs(out[0][1] - out[1][1]).max():.3e}", flush=True)
