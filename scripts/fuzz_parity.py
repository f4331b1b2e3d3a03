"""Randomised differential sweep: the GPU solver against the oracle on tanh-linear problems of random shape, bounds, starting
point and settings (run on the GPU box). Not a test of the tiers -- those pin named cases -- but a search for disagreements
outside them: every case prints one line only when something differs; the summary counts the categories.

  python scripts/fuzz_parity.py [cases=300] [seed0=0] [device|host]     host = the reference ABI: python host callbacks, optional analytic
                                                                       Jacobian and thread manager, f64 and f32, small shapes

Compared per case: status, iterations, fCalls (exact), x (1e-6 of max(1, |x|_inf)), residual (rtol 1e-6: one LM step from the same
point already differs by ~3e-8 relative -- the finite-difference Jacobian divides the 2e-16 difference of two tanh implementations by 2h = 3e-8). A case whose counters
differ while x and the residual agree is reported as `trajectory` (two roundings of one tie), anything else as `MISMATCH`."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import mir_optim_amd as M
from mir_optim_amd import workloads as W
from oracle import oracle as O
import problems as P


def case(seed):
    rng = np.random.default_rng(seed)
    # small shapes: the oracle's plain loops must finish a case in well under a second (the sweep runs on GPU-box minutes)
    n = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 96, 128, 129]))
    m = int(n + rng.integers(0, 2 * n + 50)) if rng.random() < 0.5 else int(rng.integers(max(n, 2), 1500))
    if os.environ.get("FUZZ_WIDE") == "1":              # the paths above n = 256: tile-pair J^T J, the any-n solve, the wide sweep / rewrite
        n = int(rng.choice([257, 264, 300, 320, 384, 500, 512, 513, 520]))
        m = int(n + rng.integers(0, 300)) if rng.random() < 0.6 else int(n + rng.integers(300, 1500))
    if os.environ.get("FUZZ_WIDE") == "2":              # above the read-only Broyden sweep's n = 512: J rewritten, everything through the any-n kernels
        n = int(rng.choice([527, 600, 768, 1000, 1024, 1025, 1100]))
        m = int(n + rng.integers(0, 200)) if rng.random() < 0.6 else int(n + rng.integers(200, 900))
    A = (2 * rng.random((m, n)) - 1) * np.sqrt(3.0 / n)
    xs = 2 * rng.random(n) - 1
    b = np.tanh(A @ xs) + 10.0 ** rng.integers(-6, -1) * (2 * rng.random(m) - 1)
    x0 = xs + 10.0 ** rng.integers(-3, 0) * (2 * rng.random(n) - 1)
    kind = rng.integers(0, 5)
    lo = np.full(n, -np.inf); up = np.full(n, np.inf)
    if kind >= 1:                                  # some bounds, a share of them binding at the minimiser
        sel = rng.random(n) < 0.5
        lo[sel] = xs[sel] - rng.random(sel.sum()) * 0.3 + (rng.random(sel.sum()) < 0.4) * 0.2
        sel = rng.random(n) < 0.5
        up[sel] = np.maximum(lo[sel] + 1e-3, xs[sel] + rng.random(sel.sum()) * 0.3 - (rng.random(sel.sum()) < 0.3) * 0.15)
    if kind == 3:                                  # a few pinned variables (lower == upper)
        sel = rng.random(n) < 0.15
        v = np.where(np.isfinite(lo), lo, np.where(np.isfinite(up), up, 0.1))
        lo[sel] = v[sel]; up[sel] = v[sel]
    if kind == 4:                                  # every variable boxed tightly
        lo = xs - 0.05 - 0.1 * rng.random(n); up = xs + 0.02 + 0.1 * rng.random(n)
    up = np.maximum(up, lo)
    x0 = np.clip(x0, lo, up)
    s = dict(maxIterations=int(rng.choice([1, 2, 5, 12, 40] if os.environ.get("FUZZ_WIDE") not in ("1", "2") else [1, 2, 4, 6, 12])), absTolerance=float(rng.choice([1e-3, 1e-6, 1e-9])),
             maxAge=int(rng.choice([0, 0, 1, 3])), gradTolerance=float(rng.choice([2.2e-16, 1e-8, 1e-3])))
    return dict(A=A, b=b, x0=x0, lo=lo, up=up, m=m, n=n, s=s, bounded=kind >= 1)


def host_case(seed, dtype):
    """the reference ABI: host callbacks (python), optional analytic Jacobian, optional thread manager; small shapes"""
    c = case(seed)
    rng = np.random.default_rng(seed + 77)
    n = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 33]))
    m = int(n + rng.integers(0, 60))
    c["A"] = np.ascontiguousarray(c["A"][: min(m, c["m"]), : min(n, c["n"])]) if c["n"] >= n and c["m"] >= m else (2 * rng.random((m, n)) - 1) * np.sqrt(3.0 / n)
    m, n = c["A"].shape
    xs = 2 * rng.random(n) - 1
    c["b"] = np.tanh(c["A"] @ xs) + 1e-3 * (2 * rng.random(m) - 1)
    c["lo"] = np.full(n, -np.inf); c["up"] = np.full(n, np.inf)
    if rng.random() < 0.6:
        sel = rng.random(n) < 0.5
        c["lo"][sel] = xs[sel] - 0.2 * rng.random(sel.sum()) + 0.15 * (rng.random(sel.sum()) < 0.4)
        c["up"] = np.maximum(c["lo"], np.where(rng.random(n) < 0.5, xs + 0.2 * rng.random(n) - 0.1 * (rng.random(n) < 0.3), np.inf))
        c["bounded"] = True
    else:
        c["bounded"] = False
    c["x0"] = np.clip(xs + 0.1 * (2 * rng.random(n) - 1), c["lo"], c["up"])
    c["m"], c["n"] = m, n
    c["analytic"] = bool(rng.random() < 0.4)
    c["tm"] = bool(rng.random() < 0.5)
    if dtype == np.float32:
        c["s"]["absTolerance"] = max(c["s"]["absTolerance"], 1e-6)
        c["s"]["gradTolerance"] = max(c["s"]["gradTolerance"], 1e-7)
    return c


def run_host(c, dtype, sg, so):
    A = c["A"].astype(dtype); b = c["b"].astype(dtype)

    def f(x, y):
        y[:] = np.tanh(A @ x) - b

    def g(x, J):
        J[:, :] = (1 - np.tanh(A @ x) ** 2)[:, None] * A

    def tm(count, task):                       # a thread manager that runs the tasks in reverse order on "4 threads"
        for i in reversed(range(count)):
            task(4, i % 4, i)
    lo = c["lo"].astype(dtype) if c["bounded"] else None
    up = c["up"].astype(dtype) if c["bounded"] else None
    res, x = M.optimizeLeastSquares(f, c["m"], c["x0"].astype(dtype), lo, up, g=g if c["analytic"] else None, tm=tm if c["tm"] else None,
                                    settings=sg, dtype=dtype)
    ro, xo = O.optimize(f, c["m"], c["x0"].astype(dtype), lower=lo, upper=up, g=g if c["analytic"] else None, settings=so, dtype=dtype)
    return res, x, ro, xo


def run_shards(c, sg, world):
    """the row-sharded solve on `world` in-process shards (one host thread each, all on this GPU): every rank's result"""
    import threading
    from mir_optim_amd import parallel as PAR
    comms, close = PAR.local_group(world)
    probs = []
    for r in range(world):
        off, ml = PAR.row_shard(c["m"], world, r)
        probs.append(W.TanhLinear(c["A"][off:off + ml], c["b"][off:off + ml]))
    lo = c["lo"] if c["bounded"] else None
    up = c["up"] if c["bounded"] else None
    out, err = [None] * world, [None] * world

    def one(r):
        try:
            out[r] = probs[r].solve(c["x0"], lo, up, settings=sg, comm=comms[r], batched=True)
        except BaseException as e:      # noqa: BLE001
            err[r] = e
    ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    hung = any(t.is_alive() for t in ts)
    close()
    for pb in probs:
        pb.dA.free(); pb.db.free()
    if hung or any(err):
        raise RuntimeError(f"sharded solve failed: hung={hung} {err}")
    return out


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    mode = sys.argv[3] if len(sys.argv) > 3 else "device"       # device | host (reference ABI, f64 + f32) | shards (2 .. 8 in-process row shards)
    tally = {"same": 0, "trajectory": 0, "MISMATCH": 0}
    # (no OpenBLAS in this process: its thread pool and the GPU runtime do not share a process well -- tests/test_gpu_fullsize.py runs it in one of its own)
    wide_blas = False
    wide = os.environ.get("FUZZ_WIDE") in ("1", "2")
    if wide:
        O.lib().lmo_set_omp_threads(8)      # the oracle's tanh-linear residual goes OpenMP above 200 000 elements; a GPU box shows every host core but shares 16
    for k in range(cases):
        dtype = np.float32 if (mode == "host" and k % 3 == 2) else np.float64
        c = host_case(seed0 + k, dtype) if mode == "host" else case(seed0 + k)
        sg = M.LeastSquaresSettings(dtype); so = O.default_settings(dtype)
        for key, v in c["s"].items():
            setattr(sg, key, v); setattr(so, key, v)
        lo = c["lo"] if c["bounded"] else None
        up = c["up"] if c["bounded"] else None
        if mode == "shards":
            rng = np.random.default_rng(seed0 + k + 12345)
            world = int(rng.choice([2, 3, 4, 8]))
            if c["m"] < 2 * world:
                tally["same"] += 1
                continue
            outs = run_shards(c, sg, world)
            res, x = outs[0]
            for r in range(1, world):            # every rank returns the same bits
                rr, xr = outs[r]
                if not (np.array_equal(xr, x) and rr.residual == res.residual and (int(rr.status), rr.iterations, rr.fCalls) == (int(res.status), res.iterations, res.fCalls)):
                    print(f"MISMATCH   seed {seed0 + k} shards: rank {r} of {world} differs from rank 0", flush=True)
                    tally["MISMATCH"] += 1
            xtol, rtol = 1e-6, 1e-6
            ctx = O.TanhLinearCtx(c["A"].ctypes.data, c["b"].ctypes.data)
            ro, xo = O.optimize(O.native_fn("wlc_tanh_linear_f"), c["m"], c["x0"], lower=lo, upper=up, settings=so, fctx=C.addressof(ctx))
        elif mode == "host":
            res, x, ro, xo = run_host(c, dtype, sg, so)
            xtol, rtol = (1e-6, 1e-6) if dtype == np.float64 else (5e-3, 5e-3)
            x = np.asarray(x, dtype=np.float64); xo = np.asarray(xo, dtype=np.float64)
        else:
            xtol, rtol = 1e-6, 1e-6
            prob = W.TanhLinear(c["A"], c["b"])
            res, x = prob.solve(c["x0"], lo, up, settings=sg, batched=bool(k % 2))
            prob.dA.free(); prob.db.free()
            ctx = O.TanhLinearCtx(c["A"].ctypes.data, c["b"].ctypes.data)
            ro, xo = O.optimize(O.native_fn("wlc_tanh_linear_f"), c["m"], c["x0"], lower=lo, upper=up, settings=so, fctx=C.addressof(ctx),
                                use_openblas=wide_blas)
            if wide:
                print(f"... case {k} done (m {c['m']} n {c['n']})", flush=True)      # wide cases take seconds each: keep the log moving
        scale = max(1.0, float(np.abs(xo).max()))
        xerr = float(np.abs(x - xo).max()) / scale
        rerr = abs(res.residual - ro.residual) / max(abs(ro.residual), 1e-300)
        counters = (int(res.status), res.iterations, res.fCalls) == (ro.status, ro.iterations, ro.fCalls)
        close = xerr <= xtol and (rerr <= rtol or abs(res.residual - ro.residual) <= 1e-18)
        cat = "same" if (counters and close) else ("trajectory" if close else "MISMATCH")
        tally[cat] += 1
        if cat != "same":
            print(f"{cat:10s} seed {seed0 + k} {mode} {np.dtype(dtype).name} m {c['m']} n {c['n']} bounded {c['bounded']} g {c.get('analytic')} tm {c.get('tm')} {c['s']}  gpu ({int(res.status)}, {res.iterations}, {res.fCalls}, "
                  f"{res.residual:.17g})  oracle ({ro.status}, {ro.iterations}, {ro.fCalls}, {ro.residual:.17g})  xerr {xerr:.2e} rerr {rerr:.2e}", flush=True)
        elif k % 50 == 0:
            print(f"... case {k}: ok (m {c['m']} n {c['n']})", flush=True)
    print("summary:", tally)


if __name__ == "__main__":
    main()
