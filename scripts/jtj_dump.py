"""Helper for the v2-vs-v3 bitwise test: run one fused Broyden J^T J on seeded inputs and save the outputs.
usage: python scripts/jtj_dump.py m n out.npz   (MIR_LSQ_JTJ_SPLIT=1 selects k_jtj3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M

m, n, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(m + n)
J = rng.standard_normal((m, n)); y = rng.standard_normal(m); yo = y + 0.01 * rng.standard_normal(m)
dx = 1e-2 * rng.standard_normal(n)
JJ, Jy, Jn, ms = M.jtj(J, y, yo, dx)
np.savez(out, JJ=JJ, Jy=Jy, Jn=Jn)
