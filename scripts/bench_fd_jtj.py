"""Micro-benchmark of the J^T J kernel with the finite-difference fill fused in (mir_lsq_fd_jtj_d), m x n = 1e6 x 128."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import api

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(0)
Y = rng.standard_normal((m, 2 * n))
dY = api.DeviceBuffer(Y); del Y
dt = api.DeviceBuffer(np.full(n, 2.0 ** -25)); dy = api.DeviceBuffer(rng.standard_normal(m))
dJ = api.DeviceBuffer(nbytes=m * n * 8, dtype=np.float64, shape=(m, n))
dJJ = api.DeviceBuffer(nbytes=n * n * 8, dtype=np.float64, shape=(n, n)); dJy = api.DeviceBuffer(nbytes=n * 8, dtype=np.float64, shape=(n,))
ms = C.c_float(0); ts = []
for rep in range(12):
    rc = api.lib().mir_lsq_fd_jtj_d(m, n, dY.ptr, dt.ptr, dy.ptr, dJ.ptr, dJJ.ptr, dJy.ptr, None, C.byref(ms))
    assert rc == 0
    ts.append(ms.value)
ts = sorted(ts[2:])
b = 8.0 * (3.0 * m * n + m)
print(f"fd+jtj m={m} n={n}: median {ts[len(ts)//2]:.4f} ms  min {ts[0]:.4f} ms  -> {b / ts[len(ts)//2] / 1e6:.0f} GB/s")
