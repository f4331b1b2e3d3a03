"""Generates tests/golden/lm_cases.json with the ORACLE (oracle/liblm_oracle.so, the CPU restatement
of /root/reference/source/mir/optim/least_squares.d:877-1176 + boxcqp.d:122-379).

The D reference cannot be built or imported in this image (no D compiler), so these vectors are NOT
outputs of the reference itself: they freeze the oracle, which is pinned on the reference's own
known-answer unittests (T1-T6, TQ; the `expect`/`tol` fields below are the reference's assertions,
LS:244, 272, 317, 329-330, 362, 393, 407, 433, QP:401).

    python tests/golden/make_golden.py        # rewrites lm_cases.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import problems as P  # noqa: E402
from oracle import oracle as O  # noqa: E402


def rec(res, x, extra=None):
    d = dict(status=O.STATUS[res.status], iterations=res.iterations, fCalls=res.fCalls, gCalls=res.gCalls,
             residual=float(res.residual), lambda_=float(res.lambda_), x=[float(v) for v in x])
    if extra:
        d.update(extra)
    return d


def main():
    out = {"_generator": "tests/golden/make_golden.py (oracle; reference unbuildable here)", "kats": {}, "synthetic": []}
    for name, p in (("T1", P.t1()), ("T2", P.t2()), ("T3a", P.t3a()), ("T3b", P.t3b()), ("T4", P.t4()), ("T6", P.t6())):
        res, x = O.optimize(p["f"], p["m"], p["x0"], lower=p["lower"], upper=p["upper"], g=p["g"])
        out["kats"][name] = rec(res, x, dict(expect=p.get("expect"), tol=p.get("tol")))
    a, b = P.t5()
    for name, p in (("T5a", a), ("T5b", b)):
        res, x = O.optimize(p["f"], p["m"], p["x0"], lower=p["lower"], upper=p["upper"])
        out["kats"][name] = rec(res, x)
    q = P.tq()
    st, x, it = O.solve_box_qp(q["P"], q["q"], q["l"], q["u"])
    out["kats"]["TQ"] = dict(status=int(st), iterations=it, x=[float(v) for v in x], expect=q["expect"])

    # synthetic tanh-linear family, unbounded and bounded
    for m, n, bounded in [(64, 4, False), (512, 8, False), (4096, 16, False), (20000, 32, False), (3000, 16, True)]:
        w = P.tanh_linear(m, n)
        lo = up = None
        x0 = w["x0"]
        if bounded:
            lo = w["xstar"] - 0.02
            lo[::3] = w["xstar"][::3] + 0.01
            up = w["xstar"] + 0.5
            x0 = np.clip(x0, lo, up)
        s = O.default_settings(); s.absTolerance = 1e-9
        ctx = O.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
        res, x = O.optimize(O.native_fn("wlc_tanh_linear_f"), m, x0, lower=lo, upper=up, settings=s, fctx=C.addressof(ctx))
        out["synthetic"].append(rec(res, x, dict(family="tanh_linear", m=m, n=n, bounded=bounded, absTolerance=1e-9)))
    with open(os.path.join(HERE, "lm_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "lm_cases.json"))

    # float ?posvx('E','L') with fused multiply-adds (lmo_posvx_fused_s): the bit patterns the device's one-row-per-lane solve
    # (posvx_rows) must reproduce. Matrices, right-hand sides and solutions as uint32 bit patterns of the floats.
    rng = np.random.default_rng(2026)
    systems = []
    for n in (3, 8):
        for k in range(12):
            G = rng.standard_normal((2 * n, n))
            if k % 2:
                G = G * np.logspace(-2, 2, n)[None, :]
            A = G.T @ G + 1e-3 * np.eye(n)
            if k == 7:
                A[n // 2, n // 2] = -abs(A[n // 2, n // 2])
            A = A.astype(np.float32)
            b = rng.standard_normal(n).astype(np.float32)
            info, x, eq = O.posvx_fused_s(A, b)
            systems.append(dict(n=n, A=[int(v) for v in A.view(np.uint32).ravel()], b=[int(v) for v in b.view(np.uint32)],
                                info=int(info), equilibrated=bool(eq), x=[int(v) for v in x.view(np.uint32)]))
    with open(os.path.join(HERE, "posvx_fused_s.json"), "w") as f:
        json.dump({"_generator": "tests/golden/make_golden.py: oracle lmo_posvx_fused_s", "systems": systems}, f)
    print("wrote", os.path.join(HERE, "posvx_fused_s.json"))


if __name__ == "__main__":
    main()
