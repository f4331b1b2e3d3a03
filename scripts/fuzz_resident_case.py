"""One case of scripts/fuzz_resident.py in detail: resident path, launch-chain path (tanh32 only) and oracle, with the traces of the
resident path and the oracle compared event by event. usage: python scripts/fuzz_resident_case.py <seed>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np

import mir_optim_amd as M
from mir_optim_amd import workloads as W
from oracle import oracle as O
import fuzz_resident as F

seed = int(sys.argv[1])
c = F.case(seed)
sg = M.LeastSquaresSettings(); so = O.default_settings()
for key, v in c["s"].items():
    setattr(sg, key, v); setattr(so, key, v)
print(c["model"], "m", c["m"], c["s"], "bounded", c["bounded"], "wgs", c["wgs"])
r = W.Resident(c["model"], c["rows"], max_workgroups=c["wgs"])
tr = M.Trace(8192)
res, x, st = r.solve(c["x0"], c["lo"], c["up"], settings=sg, trace=tr)
print("resident:", res, "look-ahead rejections", st["lookahead_rejections"])
res_n, x_n, st_n = r.solve(c["x0"], c["lo"], c["up"], settings=sg, variant=W.RESIDENT_NO_LOOKAHEAD)
print("resident, no look-ahead:", res_n, "same bits" if (np.array_equal(x, x_n) and res.residual == res_n.residual and res.fCalls == res_n.fCalls) else "DIFFERENT")
ev = []
ro, xo = O.optimize(O.native_fn(c["ofn"]), c["m"], c["x0"], lower=c["lo"], upper=c["up"], settings=so, fctx=C.addressof(c["octx"]), trace=lambda *a: ev.append(a))
print("oracle  :", ro.status, ro.iterations, ro.fCalls, repr(ro.residual))
if c["model"] == "tanh32":
    prob = W.TanhLinear(c["keep"][0], c["keep"][1])
    r2, x2 = prob.solve(c["x0"], c["lo"], c["up"], settings=sg)
    print("chain   :", r2, " |x_chain - x_oracle|", np.abs(x2 - xo).max())
print("|x - xo|", np.abs(x - xo).max(), "on bounds (gpu / oracle):", ((x == c["lo"]) | (x == c["up"])).sum(), ((xo == c["lo"]) | (xo == c["up"])).sum())
got = tr.records()
for k in range(min(len(got), len(ev))):
    g, e = got[k], ev[k]
    same = (int(g[0]), int(g[1])) == (int(e[0]), int(e[1])) and np.isclose(g[2], e[2], rtol=1e-6) and np.allclose(g[3:5], e[3:5], rtol=1e-7, atol=1e-300)
    if not same:
        print("first trace difference at event", k, "of", len(got), "/", len(ev))
        for j in range(max(0, k - 2), min(k + 3, len(got), len(ev))):
            print("   ", j, got[j], "|", ev[j])
        break
else:
    print("traces agree on", min(len(got), len(ev)), "events; lengths", len(got), len(ev))
