"""Micro-benchmark of the fused [Broyden +] J^T J + J^T y kernel (run on the GPU box).
usage: python scripts/bench_jtj.py [m] [n] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import api

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(0)
J = rng.standard_normal((m, n)); y = rng.standard_normal(m); yo = y + 0.01 * rng.standard_normal(m)
dx = 1e-3 * rng.standard_normal(n)
dJ, dy, dyo, ddx = api.DeviceBuffer(J), api.DeviceBuffer(y), api.DeviceBuffer(yo), api.DeviceBuffer(dx)
dJJ = api.DeviceBuffer(nbytes=n * n * 8, dtype=np.float64, shape=(n, n)); dJy = api.DeviceBuffer(nbytes=n * 8, dtype=np.float64, shape=(n,))
L = api.lib()
for br in (0, 1):
    ts = []
    for _ in range(reps):
        ms = C.c_float(0)
        rc = L.mir_lsq_jtj_d(m, n, dJ.ptr, dy.ptr, dyo.ptr, ddx.ptr, br, dJJ.ptr, dJy.ptr, None, C.byref(ms))
        assert rc == 0
        ts.append(ms.value)
    ts = np.array(ts[2:])
    byt = 8.0 * ((2 if br else 1) * m * n + (3 if br else 1) * m)
    fl = m * n * (n + 1.0) + 2.0 * m * n + (4.0 * m * n if br else 0)
    print(f"m={m} n={n} broyden={br}: median {np.median(ts):.3f} ms  min {ts.min():.3f} ms  -> {byt / np.median(ts) / 1e6:.0f} GB/s algorithmic, {fl / np.median(ts) / 1e9:.1f} TFLOP/s")
