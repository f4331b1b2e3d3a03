"""Phase stamps of k_lm_solve (VARIANT_DEBUG_SOLVE): one small device-callback solve per n, stamps on stderr.
usage: python scripts/solve_phases.py [one] [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P

ONE = "one" in sys.argv[1:]          # "one": the any-n solve without its helper workgroups (n > 256)
for n in [int(a) for a in sys.argv[1:] if a != "one"] or [128, 64, 32, 16]:
    w = P.tanh_linear(20000, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-5
    st = M.Stats()
    print(f"---- n = {n}", file=sys.stderr, flush=True)
    res, x = prob.solve(w["x0"], settings=s, batched=True, stats=st, flags=M.TIME_KERNELS, variant=M.VARIANT_DEBUG_SOLVE | (M.VARIANT_SOLVE_ONE_WORKGROUP if ONE else 0))
    print(f"n={n} {res} solve kernel avg {st.solve_ms / max(1, st.solve_launches) * 1e3:.1f} us over {st.solve_launches}", file=sys.stderr, flush=True)
