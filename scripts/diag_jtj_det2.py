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
dJ = api.DeviceBuffer(J0)
st = api.Stream()
ref = None; bad = 0
for rep in range(200):
    ms = C.c_float(0)
    assert L.mir_lsq_jtj_d(m, n, dJ.ptr, dy.ptr, dyo.ptr, ddx.ptr, 0, dJJ.ptr, dJy.ptr, st.handle, C.byref(ms)) == 0
    out = (dJJ.download(), dJy.download())
    if ref is None: ref = out
    else:
        d = [np.max(np.abs(a - b)) for a, b in zip(out, ref)]
        if any(v != 0 for v in d): bad += 1; print("  rep", rep, "diff JJ/Jy:", d)
print("non-broyden back-to-back on a stream:", bad, "of 199 differ")
