"""Pair-panel refresh kernel (fused FD -> J, J^T J, J^T y) at 128 < n <= 256, multiples of 32 and not (m = 400 000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
m = 400_000
rng = np.random.default_rng(0)
for n in (256, 250, 224, 208, 200, 192, 161, 160, 129):
    Y = rng.standard_normal((m, 2 * n)); y = rng.standard_normal(m)
    twh = np.full(n, 2.0 ** -25)
    t = min(M.fd_jtj(Y, twh, y)[3] for _ in range(3))
    print(f"n = {n:4d}: pair panel -> J, J^T J {t:.3f} ms ({8 * m * (3 * n + 1) / t / 1e6:.0f} GB/s, {m * n * (n + 3.0) / t / 1e9:.1f} TFLOP/s)", flush=True)
