"""Resident-J solver against the launch-chain path and the oracle on the small workloads (a development aid; the tests proper
are tests/test_gpu_resident.py). Usage: python scripts/resident_check.py [quick]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mir_optim_amd as M  # noqa: E402
from mir_optim_amd import workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402
import problems as P  # noqa: E402


def show(tag, res, x, st):
    print(f"{tag}: {res}", flush=True)
    if st:
        tt = st["t_total"] / 100.0
        print(f"   rounds {st['rounds']} passes {st['passes']} acc {st['accepted']} rej {st['rejected']} fd {st['jacobian_full']} "
              f"broyden {st['jacobian_broyden']} qp {st['qp_active_set_passes']} elided {st['elided_evaluations']} look-ahead {st['lookahead_rejections']} ({st['t_look']/100:.1f} us) abort {st['abort_code']} "
              f"grid {st['grid']} rows {st['rows']}", flush=True)
        print(f"   us: total {tt:.1f} stage {st['t_stage']/100:.1f} worker {st['t_worker']/100:.1f} group {st['t_group']/100:.1f} "
              f"total-wait {st['t_total_wait']/100:.1f} solver {st['t_solver']/100:.1f} (solve body {st['t_solve_body']/100:.1f}) "
              f"cmd-wait {st['t_cmd_wait']/100:.1f}  per round {tt / max(1, st['rounds']):.2f}  [worker: eval {st['t_w_eval']/100:.1f} fd {st['t_w_fd']/100:.1f} "
              f"products+publish {st['t_w_prod']/100:.1f} (matrix-core loop {st['t_w_mma']/100:.1f})]", flush=True)


def gauss(m, K, bounded=None):
    g = P.gauss_sum(m, K=K)
    lower, upper, x0 = g["lower"], g["upper"], g["x0"]
    if bounded:
        lower, upper = lower.copy(), upper.copy()
        lower[2 * K] = 0.045; lower[2 * K + min(3, K - 1)] = 0.05
        upper[0] = 0.95; upper[min(3, K - 1)] = 0.85
        x0 = np.clip(x0, lower, upper)
    r = W.Resident.gauss_sum(g["t"], g["data"], K=K)
    tr = M.Trace(4096)
    res, x, st = r.solve(x0, lower, upper, trace=tr)
    show(f"resident gauss m={m} K={K} bounded={bool(bounded)}", res, x, st)
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    tr2 = M.Trace(4096)
    res2, x2 = prob.solve(x0, lower, upper, trace=tr2)
    print(f"   chain   : {res2}")
    ctx = O.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
    ev = []
    ro, xo = O.optimize(O.native_fn("wlc_gauss_sum_f"), g["m"], x0, lower=lower, upper=upper, fctx=C.addressof(ctx),
                        trace=lambda *a: ev.append(a))
    print(f"   oracle  : status {ro.status} it {ro.iterations} fCalls {ro.fCalls} residual {ro.residual!r} lambda {ro.lambda_!r}")
    print(f"   |x - xo|max {np.abs(x - xo).max():.3e}  |x2 - xo|max {np.abs(x2 - xo).max():.3e}  trace events res {tr.count} chain {tr2.count} oracle {len(ev)}")
    got = tr.records()
    nshow = 0
    for k in range(min(len(got), len(ev))):
        g_, e_ = got[k], ev[k]
        same = (int(g_[0]), int(g_[1])) == (int(e_[0]), int(e_[1])) and np.isclose(g_[2], e_[2], rtol=1e-6) and np.allclose(g_[3:5], e_[3:5], rtol=1e-9, atol=1e-300)
        if not same:
            print(f"   first trace difference at event {k}: resident {g_} oracle {e_}")
            nshow = 1
            break
    if not nshow:
        print(f"   traces agree on the first {min(len(got), len(ev))} events")
    # timing: repeated solves
    ts = []
    for _ in range(5):
        r.upload_point(x0, lower, upper)
        t0 = time.perf_counter()
        r.launch()
        r.stream.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"   host wall per solve: {min(ts) * 1e3:.3f} ms (min of 5)")


def tanh32(m):
    w = P.tanh_linear(m, 32)
    r = W.Resident.tanh_linear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    tr = M.Trace(4096)
    res, x, st = r.solve(w["x0"], settings=s, trace=tr, variant=W.RESIDENT_UNBOUNDED)
    show(f"resident tanh32 m={m}", res, x, st)
    so = O.default_settings(); so.absTolerance = 1e-9
    ctx = O.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ev = []
    ro, xo = O.optimize(O.native_fn("wlc_tanh_linear_f"), w["m"], w["x0"], settings=so, fctx=C.addressof(ctx), trace=lambda *a: ev.append(a))
    print(f"   oracle  : status {ro.status} it {ro.iterations} fCalls {ro.fCalls} residual {ro.residual!r}")
    print(f"   |x - xo|max {np.abs(x - xo).max():.3e} events {tr.count} / {len(ev)}")


if __name__ == "__main__":
    gauss(20000, 3)
    tanh32(20000)
    gauss(100000, 5)
    gauss(100000, 5, bounded=True)
