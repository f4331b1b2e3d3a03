"""Randomised differential sweep of the standalone BOXCQP entry (mir_solve_box_qp_gpu_d / _s) against the oracle's solveBoxQP
(boxcqp.d:122-379): random SPD matrices of every size class of the solve kernels (one LDS block ... the global-memory factor ... the
any-n path above 256), conditioning from benign to equilibration-triggering, bounds from absent to tight / pinned / infeasible
starts. Prints only the cases that differ.   python scripts/fuzz_boxqp.py [cases=300] [seed0=0]   (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import mir_optim_amd as M
from mir_optim_amd import api
from oracle import oracle as O


def case(seed, dtype):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 160, 192, 255, 256, 257, 300]))
    if dtype == np.float32:
        n = min(n, 64)
    G = rng.standard_normal((n + int(rng.integers(0, 2 * n + 3)), n))
    P = G.T @ G / G.shape[0] + 10.0 ** rng.integers(-6, 0) * np.eye(n)
    if rng.random() < 0.4:                                     # badly scaled rows / columns: ?poequ + ?laqsy
        d = 10.0 ** rng.uniform(-3, 3, n) if dtype == np.float64 else 10.0 ** rng.uniform(-1.5, 1.5, n)
        P = P * d[:, None] * d[None, :]
    q = rng.standard_normal(n) * 10.0 ** rng.integers(-2, 2)
    xu = -np.linalg.solve(P, q)
    kind = rng.integers(0, 5)
    l = np.full(n, -np.inf); u = np.full(n, np.inf)
    if kind >= 1:
        sel = rng.random(n) < 0.6
        l[sel] = xu[sel] - np.abs(xu[sel]) * rng.random(sel.sum()) + (rng.random(sel.sum()) < 0.5) * np.abs(xu[sel]) * 0.7
        sel = rng.random(n) < 0.6
        u[sel] = np.maximum(l[sel], xu[sel] + np.abs(xu[sel]) * rng.random(sel.sum()) - (rng.random(sel.sum()) < 0.5) * np.abs(xu[sel]) * 0.7)
    if kind == 3:
        sel = rng.random(n) < 0.2
        v = np.where(np.isfinite(l), l, 0.0)
        l[sel] = v[sel]; u[sel] = v[sel]
    if kind == 4:
        l = xu - 1e-3 * np.abs(xu) * rng.random(n) - 1e-9; u = xu + 1e-3 * np.abs(xu) * rng.random(n) + 1e-9
    u = np.maximum(u, l)
    return P.astype(dtype), q.astype(dtype), l.astype(dtype), u.astype(dtype)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tally = {"same": 0, "iterations": 0, "MISMATCH": 0}
    for k in range(cases):
        dtype = np.float32 if k % 5 == 4 else np.float64
        P, q, l, u = case(seed0 + k, dtype)
        st, x, it = api.solveBoxQP(P, q, l, u, dtype=dtype)
        so, xo, ito = O.solve_box_qp(P, q, l, u, dtype=dtype)
        tol = 1e-9 if dtype == np.float64 else 2e-4
        scale = max(1.0, float(np.abs(xo).max()))
        xerr = float(np.abs(x.astype(np.float64) - xo.astype(np.float64)).max()) / scale if int(st) == 0 and so == 0 else 0.0
        ok_status = int(st) == so
        cat = "same" if (ok_status and it == ito and xerr <= tol) else ("iterations" if (ok_status and xerr <= tol) else "MISMATCH")
        tally[cat] += 1
        if cat != "same":
            print(f"{cat:10s} seed {seed0 + k} n {P.shape[0]} {np.dtype(dtype).name}  gpu (status {int(st)}, it {it})  oracle ({so}, {ito})  xerr {xerr:.2e}", flush=True)
    print("summary:", tally)


if __name__ == "__main__":
    main()
