"""Dump the bits of a cfg 5 (pad8) run of the library at argv[1] to argv[2] (.npz): A/B of kernel variants."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import mir_optim_amd.build as B
if len(sys.argv) > 1 and sys.argv[1] != "-":
    B.SOLVER_LIB = os.path.abspath(sys.argv[1])
    B.build = lambda *a, **k: (B.SOLVER_LIB, B.WORKLOADS_LIB)
import mir_optim_amd as M
from mir_optim_amd import api
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import problems as P

count, m, n = 4096, 512, 8
t, data, truth, x0 = P.cfg5_pad8(count, m)
res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP_DECAY_PAD8, x0, t, data, settings=M.LeastSquaresSettings(np.float32))
np.savez(sys.argv[2], x=x, it=np.array([r.iterations for r in res]), st=np.array([int(r.status) for r in res]),
         f=np.array([r.fCalls for r in res]), r=np.array([r.residual for r in res], dtype=np.float32), g=np.array([r.gCalls for r in res]))
print("wrote", sys.argv[2], "iterations", sum(r.iterations for r in res), "fCalls", sum(r.fCalls for r in res), "gCalls", sum(r.gCalls for r in res))
