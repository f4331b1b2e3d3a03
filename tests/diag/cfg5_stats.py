import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
import test_gpu_batched as T
t, data, truth, x0 = T.make_exp3(4096)
res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP3_AFFINE, x0, t, data)
it = np.array([r.iterations for r in res]); fc = np.array([r.fCalls for r in res]); st = np.array([int(r.status) for r in res])
print("iterations percentiles 50/90/99/max:", np.percentile(it, [50, 90, 99]), it.max())
print("fCalls percentiles 50/90/99/max:", np.percentile(fc, [50, 90, 99]), fc.max())
for s in np.unique(st):
    print("status", s, "count", (st == s).sum(), "mean it", it[st == s].mean(), "max it", it[st == s].max(), "mean fcalls", fc[st == s].mean())
