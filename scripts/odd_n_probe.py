"""Kernel times of the plain J^T J and of the difference-panel refresh at odd / even n (unit entries; m = 1e6).
usage (GPU box): python scripts/odd_n_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M

m = 1_000_000
rng = np.random.default_rng(0)
for n in (128, 127, 126, 100, 99, 65, 64):
    J = rng.standard_normal((m, n)); y = rng.standard_normal(m)
    twh = np.full(n, 2.0 ** -25)
    tp = min(M.jtj(J, y)[3] for _ in range(3))
    td = min(M.fd_jtj(J, twh, y, diff=True)[3] for _ in range(3))
    print(f"n = {n:4d}: plain J^T J {tp:.3f} ms ({8 * m * (n + 1) / tp / 1e6:.0f} GB/s)   difference panel -> J, J^T J {td:.3f} ms "
          f"({8 * m * (2 * n + 1) / td / 1e6:.0f} GB/s)", flush=True)
