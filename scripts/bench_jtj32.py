"""Micro-benchmark of the f32 J^T J + J^T y kernel (mir_lsq_jtj_s: k_jtj_pc32, jtj_pc32.h) against numpy.
usage: python scripts/bench_jtj32.py [m] [n]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import api

m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
for n in ([int(a) for a in sys.argv[2:]] or [128, 64, 100, 32]):
    rng = np.random.default_rng(n)
    J = rng.standard_normal((m, n)).astype(np.float32); y = rng.standard_normal(m).astype(np.float32)
    dJ, dy = api.DeviceBuffer(J), api.DeviceBuffer(y)
    dJJ = api.DeviceBuffer(nbytes=n * n * 4, dtype=np.float32, shape=(n, n)); dJy = api.DeviceBuffer(nbytes=n * 4, dtype=np.float32, shape=(n,))
    L = api.lib()
    L.mir_lsq_jtj_s.restype = C.c_int
    ts = []
    for _ in range(12):
        ms = C.c_float(0)
        rc = L.mir_lsq_jtj_s(C.c_size_t(m), C.c_size_t(n), C.c_void_p(dJ.ptr), C.c_void_p(dy.ptr), C.c_void_p(dy.ptr), C.c_void_p(dy.ptr), 0,
                             C.c_void_p(dJJ.ptr), C.c_void_p(dJy.ptr), None, C.byref(ms))
        assert rc == 0, rc
        ts.append(ms.value)
    ts = sorted(ts[2:]); med = ts[len(ts) // 2]
    JJ, Jy = dJJ.download(), dJy.download()
    sub = slice(0, min(m, 200000))
    refJJ = J.astype(np.float64).T @ J.astype(np.float64); refJy = J.astype(np.float64).T @ y.astype(np.float64)
    err = np.abs(JJ - refJJ).max() / np.abs(refJJ).max(); erry = np.abs(Jy - refJy).max() / np.abs(refJy).max()
    by = 4.0 * (m * n + m); fl = m * n * (n + 1.0) + 2.0 * m * n
    print(f"f32 jtj m={m} n={n}: median {med:.4f} ms  -> {by / med / 1e6:.0f} GB/s ({by / med / 1e6 / 8000:.2f} of HBM), {fl / med / 1e9:.1f} TFLOP/s ({fl / med / 1e9 / 157.3:.2f} of f32 MFMA)   rel err JJ {err:.1e} Jy {erry:.1e}")
    dJ.free(); dy.free()
