import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mir_optim_amd import api
m, n = int(os.environ.get("M", 200000)), int(os.environ.get("N", 128))
rng = np.random.default_rng(0)
J0 = rng.standard_normal((m, n)); y = rng.standard_normal(m); yo = y + 0.01 * rng.standard_normal(m); dx = 1e-3 * rng.standard_normal(n)
L = api.lib()
dy, dyo, ddx = api.DeviceBuffer(y), api.DeviceBuffer(yo), api.DeviceBuffer(dx)
dJJ = api.DeviceBuffer(nbytes=n * n * 8, dtype=np.float64, shape=(n, n)); dJy = api.DeviceBuffer(nbytes=n * 8, dtype=np.float64, shape=(n,))
for br in (0, 1):
    ref = None; bad = 0
    for rep in range(12):
        dJ = api.DeviceBuffer(J0)
        ms = C.c_float(0)
        assert L.mir_lsq_jtj_d(m, n, dJ.ptr, dy.ptr, dyo.ptr, ddx.ptr, br, dJJ.ptr, dJy.ptr, None, C.byref(ms)) == 0
        out = (dJJ.download(), dJy.download(), dJ.download()); dJ.free()
        if ref is None: ref = out
        else:
            d = [np.max(np.abs(a - b)) for a, b in zip(out, ref)]
            if any(v != 0 for v in d): bad += 1; print("  rep", rep, "diff JJ/Jy/J:", d)
    Jl = ref[2].astype(np.longdouble)
    print(f"broyden={br}: {bad} of 11 repeats differ; err vs longdouble: {np.max(np.abs(ref[0] - np.asarray(Jl.T @ Jl, dtype=np.float64))):.3e}")
