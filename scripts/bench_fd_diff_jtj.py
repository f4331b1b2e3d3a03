"""Micro-benchmark of the fused finite-difference J^T J kernel on the m x n DIFFERENCE panel (mir_lsq_fd_diff_jtj_d) and of the
plain J^T J (mir_lsq_jtj_d). usage: python scripts/bench_fd_diff_jtj.py [m] [n]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import api

m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(0)
dD = api.DeviceBuffer(rng.standard_normal((m, n)))
dt = api.DeviceBuffer(np.full(n, 2.0 ** -25)); dy = api.DeviceBuffer(rng.standard_normal(m))
dJ = api.DeviceBuffer(nbytes=m * n * 8, dtype=np.float64, shape=(m, n))
dJJ = api.DeviceBuffer(nbytes=n * n * 8, dtype=np.float64, shape=(n, n)); dJy = api.DeviceBuffer(nbytes=n * 8, dtype=np.float64, shape=(n,))
L = api.lib()
for name, call in (("fd diff + jtj", lambda ms: L.mir_lsq_fd_diff_jtj_d(m, n, dD.ptr, dt.ptr, dy.ptr, dJ.ptr, dJJ.ptr, dJy.ptr, None, C.byref(ms))),
                   ("plain jtj", lambda ms: L.mir_lsq_jtj_d(m, n, dD.ptr, dy.ptr, dy.ptr, dy.ptr, 0, dJJ.ptr, dJy.ptr, None, C.byref(ms)))):
    ts = []
    for rep in range(14):
        ms = C.c_float(0)
        assert call(ms) == 0
        ts.append(ms.value)
    ts = sorted(ts[2:])
    fl = m * n * (n + 1.0) + 2.0 * m * n
    print(f"{name:14s} m={m} n={n}: median {ts[len(ts)//2]:.4f} ms  min {ts[0]:.4f} ms  -> {fl / ts[len(ts)//2] / 1e9:.1f} TFLOP/s")
