"""Randomised differential sweep of the RESIDENT-J path (include/mir_optim_amd_resident.hpp) against the oracle: the four
compiled-in models (Gaussian sums n = 10 / 16, tanh-linear n = 32, exponential decay n = 3) on random data, starting points,
bounds (absent / binding / pinned / tight boxes), settings and GRID sizes (1 ... 256 workgroups: the slice / group / leader
arithmetic). Not a test of the tiers -- those pin named cases -- but a search for disagreements outside them.

  python scripts/fuzz_resident.py [cases=400] [seed0=0]

Compared per case: status, iterations, fCalls (exact), x (1e-6 of max(1, |x|_inf)), residual (rtol 1e-6). A case whose counters
differ while x and the residual agree is `trajectory` (two roundings of one tie), anything else `MISMATCH`. Every 25th case is
also run twice and must return the same bits (the fixed-order reductions)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import mir_optim_amd as M
from mir_optim_amd import workloads as W
from oracle import oracle as O


def bounds(rng, truth, kind, spread):
    n = truth.size
    lo = np.full(n, -np.inf); up = np.full(n, np.inf)
    if kind >= 1:
        sel = rng.random(n) < 0.5
        lo[sel] = truth[sel] - spread[sel] * (rng.random(sel.sum()) - 0.35)          # a third of them above the true value
        sel = rng.random(n) < 0.5
        up[sel] = np.maximum(lo[sel] + 1e-3 * spread[sel], truth[sel] + spread[sel] * (rng.random(sel.sum()) - 0.3))
    if kind == 3:
        sel = rng.random(n) < 0.2
        v = np.where(np.isfinite(lo), lo, np.where(np.isfinite(up), up, truth))
        lo[sel] = v[sel]; up[sel] = v[sel]
    if kind == 4:
        lo = truth - spread * (0.05 + 0.3 * rng.random(n)); up = truth + spread * (0.02 + 0.3 * rng.random(n))
    return lo, np.maximum(up, lo)


def case(seed):
    rng = np.random.default_rng(seed)
    model = str(rng.choice(["gauss3", "gauss5", "tanh32", "exp_decay1"]))
    kind = int(rng.integers(0, 5))
    if model in ("gauss3", "gauss5"):
        K = 3 if model == "gauss3" else 5
        m = int(rng.choice([40, 200, 1000, 5000, 20000, 60000])) + int(rng.integers(0, 37))
        t = np.sort(rng.random(m)) if rng.random() < 0.5 else np.arange(m) / max(1, m - 1)
        a = 0.3 + rng.random(K); c = (np.arange(K) + 0.3 + 0.4 * rng.random(K)) / K; w = (0.1 + 0.2 * rng.random(K)) / K
        truth = np.concatenate([a, c, w, [0.2 * rng.random()]])
        data = sum(a[k] * np.exp(-(t - c[k]) ** 2 / (2 * w[k] ** 2)) for k in range(K)) + truth[-1]
        data = data + 10.0 ** rng.integers(-5, -1) * (2 * rng.random(m) - 1)
        spread = np.concatenate([0.3 * a, np.full(K, 0.1 / K), 0.3 * w, [0.1]])
        x0 = truth * (1 + 0.04 * (2 * rng.random(truth.size) - 1))
        lo, up = bounds(rng, truth, kind, spread)
        lo[2 * K:3 * K] = np.maximum(lo[2 * K:3 * K], 1e-3)                          # widths stay positive
        up = np.maximum(up, lo)
        rows = np.stack([t, data], axis=1)
        keep = (np.ascontiguousarray(rows[:, 0]), np.ascontiguousarray(rows[:, 1]))
        ofn, octx = "wlc_gauss_sum_f", O.GaussSumCtx(keep[0].ctypes.data, keep[1].ctypes.data)
    elif model == "tanh32":
        n = 32
        m = int(rng.choice([64, 300, 2000, 10000, 40000])) + int(rng.integers(0, 29))
        A = (2 * rng.random((m, n)) - 1) * np.sqrt(3.0 / n)
        truth = 2 * rng.random(n) - 1
        b = np.tanh(A @ truth) + 10.0 ** rng.integers(-6, -1) * (2 * rng.random(m) - 1)
        x0 = truth + 10.0 ** rng.integers(-3, 0) * (2 * rng.random(n) - 1)
        lo, up = bounds(rng, truth, kind, np.full(n, 0.4))
        rows = np.concatenate([A, b[:, None]], axis=1)
        keep = (np.ascontiguousarray(A), np.ascontiguousarray(b))
        ofn, octx = "wlc_tanh_linear_f", O.TanhLinearCtx(keep[0].ctypes.data, keep[1].ctypes.data)
    else:
        m = int(rng.choice([20, 100, 1000, 30000])) + int(rng.integers(0, 13))
        t = np.linspace(1.0, 100.0, m)
        truth = np.array([5 + 10 * rng.random(), 5 + 15 * rng.random(), 5 + 10 * rng.random()])
        data = truth[0] * np.exp(-t / truth[1]) + truth[2] + 10.0 ** rng.integers(-4, 0) * rng.standard_normal(m)
        x0 = truth * (1 + 0.3 * (2 * rng.random(3) - 1))
        lo, up = bounds(rng, truth, kind, 0.5 * truth)
        rows = np.stack([t, data], axis=1)
        keep = (np.ascontiguousarray(t), np.ascontiguousarray(data))
        ofn, octx = "wlc_exp_decay_f", O.ExpDecayCtx(keep[0].ctypes.data, keep[1].ctypes.data, 1)
    x0 = np.clip(x0, lo, up)
    s = dict(maxIterations=int(rng.choice([1, 3, 12, 60, 1000])), absTolerance=float(rng.choice([2.2e-16, 1e-6, 1e-9])),
             maxAge=int(rng.choice([0, 0, 1, 3])), gradTolerance=float(rng.choice([2.2e-16, 1e-8, 1e-3])))
    wgs = int(rng.choice([0, 0, 1, 2, 5, 16, 17, 64, 200]))
    return dict(model=model, rows=rows, m=m, x0=x0, lo=lo, up=up, s=s, wgs=wgs, ofn=ofn, octx=octx, keep=keep, bounded=kind >= 1)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tally = {"same": 0, "trajectory": 0, "MISMATCH": 0, "does-not-fit": 0, "not-reproducible": 0}
    for k in range(cases):
        c = case(seed0 + k)
        sg = M.LeastSquaresSettings(); so = O.default_settings()
        for key, v in c["s"].items():
            setattr(sg, key, v); setattr(so, key, v)
        r = W.Resident(c["model"], c["rows"], max_workgroups=c["wgs"])
        if r.plan_rc != 0:
            tally["does-not-fit"] += 1
            continue
        res, x, st = r.solve(c["x0"], c["lo"], c["up"], settings=sg)
        if k % 25 == 0:
            res2, x2, _ = r.solve(c["x0"], c["lo"], c["up"], settings=sg)
            if not (np.array_equal(x, x2) and res.residual == res2.residual and res.fCalls == res2.fCalls):
                tally["not-reproducible"] += 1
                print(f"NOT REPRODUCIBLE seed {seed0 + k} {c['model']} m {c['m']}", flush=True)
        ro, xo = O.optimize(O.native_fn(c["ofn"]), c["m"], c["x0"], lower=c["lo"], upper=c["up"], settings=so, fctx=C.addressof(c["octx"]))
        scale = max(1.0, float(np.abs(xo).max()))
        xerr = float(np.abs(x - xo).max()) / scale
        rerr = abs(res.residual - ro.residual) / max(abs(ro.residual), 1e-300)
        counters = (int(res.status), res.iterations, res.fCalls) == (ro.status, ro.iterations, ro.fCalls)
        close = xerr <= 1e-6 and (rerr <= 1e-6 or abs(res.residual - ro.residual) <= 1e-18)
        cat = "same" if (counters and close) else ("trajectory" if close else "MISMATCH")
        tally[cat] += 1
        if st["abort_code"]:
            print(f"ABORT {st['abort_code']} seed {seed0 + k}", flush=True)
        if cat == "MISMATCH" or (cat != "same" and os.environ.get("FUZZ_VERBOSE")):
            print(f"{cat:10s} seed {seed0 + k} {c['model']} m {c['m']} grid {st['grid']} bounded {c['bounded']} {c['s']}  gpu ({int(res.status)}, {res.iterations}, "
                  f"{res.fCalls}, {res.residual:.17g})  oracle ({ro.status}, {ro.iterations}, {ro.fCalls}, {ro.residual:.17g})  xerr {xerr:.2e} rerr {rerr:.2e}",
                  flush=True)
        elif k % 10 == 0:
            print(f"... case {k}: {cat} ({c['model']} m {c['m']} grid {st['grid']})", flush=True)
        for b in (r.d_rows, r.d_ws, r.d_x, r.d_out):
            b.free()
    print("summary:", tally)


if __name__ == "__main__":
    main()
