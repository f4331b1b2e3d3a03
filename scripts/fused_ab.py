"""Fused rounds against the one-by-one rounds (MIR_LSQ_VARIANT_NO_PIPELINE) on tanh-linear problems of several n: results (must be
the same bits), fused rounds / passes, launches per round kind. usage: python scripts/fused_ab.py"""
import os, sys; sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'), os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')]
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P
for n in (16, 32, 128, 200):
    w = P.tanh_linear(30000, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    for variant in (M.VARIANT_NO_PIPELINE, 0):
        st = M.Stats()
        r, x = prob.solve(w["x0"], settings=s, stats=st, batched=True, variant=variant)
        print(n, variant, r, st.fused_rounds, st.fused_passes, list(st.round_launches), list(st.rounds), flush=True)
