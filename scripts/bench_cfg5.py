"""cfg 5 (4096 small fp32 fits, one wavefront per problem): time of the C entry itself (run on the GPU box)."""
import ctypes as C, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import api
import test_gpu_batched as T

L = api.lib()
for model, make in ((M.MODEL_EXP_DECAY, T.make_exp_decay), (M.MODEL_EXP3_AFFINE, T.make_exp3)):
    t, data, truth, x0 = make(4096)
    x0 = np.ascontiguousarray(x0, dtype=np.float32); data = np.ascontiguousarray(data, dtype=np.float32); t = np.ascontiguousarray(t, dtype=np.float32)
    count, n = x0.shape; m = data.shape[1]
    lo = np.full(n, -np.inf, dtype=np.float32); up = np.full(n, np.inf, dtype=np.float32)
    s = M.LeastSquaresSettings(np.float32)
    raw = (api._Rs * count)()
    ts = []
    for rep in range(6):
        x = x0.copy()
        t0 = time.perf_counter()
        rc = L.mir_optimize_least_squares_batched_s(C.byref(s), count, m, int(model), x.ctypes.data, lo.ctypes.data, up.ctypes.data,
                                                    t.ctypes.data, 0 if t.ndim == 1 else m, data.ctypes.data, raw)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    st = collections.Counter(int(r.status) for r in raw)
    print("model %d: C entry %.2f ms per 4096 fits (best of 5 warm; first %.2f ms) -> %.0f fits/s; statuses %s"
          % (model, min(ts[1:]) * 1e3, ts[0] * 1e3, 4096 / min(ts[1:]), dict(st)))
