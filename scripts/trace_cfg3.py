"""Diagnostic: per-pass trace of the cfg-3 solve (event, iterations, lambda, residual, trial residual, |dx|) for the
two Broyden implementations. usage: python scripts/trace_cfg3.py [m] [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mir_optim_amd as M
from mir_optim_amd import workloads as W

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
data = W.tanh_linear_data(m, n)
prob = W.TanhLinear(data["A"], data["b"])
for tol in (1e-9,):
    for mode in ("fused", "lowrank"):
        if mode == "fused":
            os.environ["MIR_LSQ_BROYDEN"] = "fused"
        else:
            os.environ.pop("MIR_LSQ_BROYDEN", None)
        s = M.LeastSquaresSettings(); s.absTolerance = tol
        tr = M.Trace(4096); st = M.Stats()
        res, x = prob.solve(data["x0"], settings=s, trace=tr, stats=st, batched=True)
        print(f"== {mode} absTolerance={tol:g}: {res} passes={st.passes} full={st.jacobian_full} broyden={st.jacobian_broyden}")
        for r in tr.records()[:40]:
            print(f"   {M.Trace.EVENTS[r[0]]:17s} it={r[1]:3d} lambda={r[2]:.3e} res={r[3]:.17g} trial={r[4]:.17g} |dx|={np.sqrt(r[5]):.3e}")
