"""Soak: repeated solves and repeated fused-FD kernel launches must be bitwise identical (fixed-order reductions, no
atomics; catches synchronisation bugs in the producer/consumer and ring kernels). usage: python scripts/soak_determinism.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for (m, n) in [(60000, 128), (50000, 64), (40002, 96), (30000, 32)]:
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    ref = None
    for r in range(reps):
        res, x = prob.solve(w["x0"], settings=s, batched=True)
        key = (res.iterations, res.fCalls, res.residual, res.lambda_, x.tobytes())
        if ref is None:
            ref = key
        elif key != ref:
            bad += 1
            print(f"MISMATCH solve m={m} n={n} rep={r}: {key[:4]} vs {ref[:4]}")
    print(f"solve m={m} n={n}: {reps} repeats ok" if bad == 0 else f"solve m={m} n={n}: {bad} mismatches so far")
rng = np.random.default_rng(1)
for (m, n) in [(200000, 128), (100000, 48)]:
    Y = rng.standard_normal((m, 2 * n)); twh = np.full(n, 2.0 ** -25); y = rng.standard_normal(m)
    ref = None
    for r in range(reps // 4):
        J, JJ, Jy, _ = M.fd_jtj(Y, twh, y)
        key = (J.tobytes(), JJ.tobytes(), Jy.tobytes())
        if ref is None:
            ref = key
        elif key != ref:
            bad += 1
            print(f"MISMATCH fd_jtj m={m} n={n} rep={r}")
    print(f"fd_jtj m={m} n={n}: {reps // 4} repeats", "ok" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
