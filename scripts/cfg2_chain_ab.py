"""cfg 2's launch chain (m = 1e5 x n = 16, bounded, device callbacks) under variant bits: wall time per fit (median of the
repetitions), rounds by kind, fused rounds, and the host's own time per category of runtime call
(MIR_LSQ_VARIANT_HOST_PROFILE, printed by the library on stderr).   python scripts/cfg2_chain_ab.py [reps=200]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import mir_optim_amd as M
from mir_optim_amd import api, workloads as W
import problems as P

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = P.gauss_sum(100000, K=5)
prob = W.Curve("gauss_sum", g["t"], g["data"])
ws = api.lib().mir_lsq_workspace_create(g["m"], g["n"], 8)
ref = None
for name, variant in (("one-by-one rounds", M.VARIANT_NO_PIPELINE), ("fused rounds", 0)):
    for _ in range(5):
        prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, variant=variant, batched=True)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, x = prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, variant=variant, batched=True)
        ts.append(time.perf_counter() - t0)
    st = M.Stats()
    prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, variant=variant, batched=True, stats=st)
    key = (x.tobytes(), res.iterations, res.fCalls, res.residual, int(res.status))
    ref = ref or key
    ts = np.sort(np.array(ts)) * 1e3
    print(f"{name:24s} {np.median(ts):.3f} ms per fit (min {ts[0]:.3f}, p90 {ts[int(0.9 * reps)]:.3f})  {res.iterations} it {st.passes} passes  rounds {list(st.rounds)}"
          f"  fused {st.fused_rounds} / passes run ahead {st.fused_passes}"
          f"  {'same bits' if key == ref else 'DIFFERENT'}", flush=True)
    sys.stderr.flush()
    prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, variant=variant | M.VARIANT_HOST_PROFILE, batched=True)
