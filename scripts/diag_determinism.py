"""Diagnostic: is a repeated solve bit-reproducible? (run on the GPU box)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import api, workloads as W
m, n = int(os.environ.get("M", 200000)), int(os.environ.get("N", 128))
batched = os.environ.get("FD", "batched") == "batched"
data = W.tanh_linear_data(m, n)
prob = W.TanhLinear(data["A"], data["b"])
s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
ws = api.lib().mir_lsq_workspace_create(m, n, 8)
for rep in range(int(os.environ.get("REPS", 8))):
    st = M.Stats()
    t0 = time.perf_counter()
    res, x = prob.solve(data["x0"], settings=s, stats=st, workspace=ws, batched=batched)
    print("%.1f ms" % ((time.perf_counter() - t0) * 1e3), rep, res.status.name, res.iterations, st.passes, res.fCalls, repr(res.residual), hashlib.md5(x.tobytes()).hexdigest()[:10])
