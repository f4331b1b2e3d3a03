"""Repeat the any-n solve with helper workgroups many times (and from two host threads at once) and compare every result bit
for bit with the first: the job hand-offs are timing dependent, the results must not be. usage: python scripts/coop_stress.py [reps]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for n, m in ((300, 2500), (520, 3000), (1024, 2600)):
    w = P.tanh_linear(m, n)
    lo = w["xstar"] - 0.5; up = w["xstar"] + 0.5
    lo[::7] = w["xstar"][::7] + 0.02
    x0 = np.clip(w["x0"], lo, up)
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-7
    probs = [W.TanhLinear(w["A"], w["b"]) for _ in range(2)]
    ref, xref = probs[0].solve(x0, lo, up, settings=s, batched=True)
    t0 = time.perf_counter()

    def work(k, out):
        for _ in range(reps):
            r, x = probs[k].solve(x0, lo, up, settings=s, batched=True)
            if not (x.tobytes() == xref.tobytes() and r.residual == ref.residual and r.fCalls == ref.fCalls and r.status == ref.status):
                out.append((k, r))
    out = []
    th = [threading.Thread(target=work, args=(k, out)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    bad += len(out)
    print(f"n={n}: {2 * reps} solves from two threads in {time.perf_counter() - t0:.1f} s, {len(out)} differ from the first  ({ref})", flush=True)
print("OK" if bad == 0 else f"{bad} RESULTS DIFFER")
