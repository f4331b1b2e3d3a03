"""One case of scripts/fuzz_parity.py's wide tiers (FUZZ_WIDE=1 / 2 in the environment) with and without the helper workgroups
of the any-n solve, against the oracle. usage: FUZZ_WIDE=2 python scripts/fuzz_wide_case.py <seed> ..."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
from oracle import oracle as O
import fuzz_parity as F

O.lib().lmo_set_omp_threads(8)
for seed in [int(a) for a in sys.argv[1:]]:
    c = F.case(seed)
    sg = M.LeastSquaresSettings(); so = O.default_settings()
    for key, v in c["s"].items():
        setattr(sg, key, v); setattr(so, key, v)
    lo = c["lo"] if c["bounded"] else None
    up = c["up"] if c["bounded"] else None
    ctx = O.TanhLinearCtx(c["A"].ctypes.data, c["b"].ctypes.data)
    ro, xo = O.optimize(O.native_fn("wlc_tanh_linear_f"), c["m"], c["x0"], lower=lo, upper=up, settings=so, fctx=C.addressof(ctx))
    print(f"seed {seed} m {c['m']} n {c['n']} {c['s']}: oracle ({ro.status}, {ro.iterations}, {ro.fCalls}, {ro.residual:.17g})")
    for variant, tag in ((0, "helpers"), (M.VARIANT_SOLVE_ONE_WORKGROUP, "one workgroup")):
        prob = W.TanhLinear(c["A"], c["b"])
        res, x = prob.solve(c["x0"], lo, up, settings=sg, batched=bool(seed % 2), variant=variant)
        print(f"   {tag:14s} ({int(res.status)}, {res.iterations}, {res.fCalls}, {res.residual:.17g})  xerr {np.abs(x - xo).max() / max(1, np.abs(xo).max()):.2e} "
              f"rerr {abs(res.residual - ro.residual) / max(abs(ro.residual), 1e-300):.2e}")
