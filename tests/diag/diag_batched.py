import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, collections
import mir_optim_amd as M
from oracle import oracle as O
import test_gpu_batched as T
count = 512
t, data, truth, x0 = T.make_exp3(count)
res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP3_AFFINE, x0, t, data)
st = np.array([int(r.status) for r in res])
print(collections.Counter(st.tolist()))
bad = np.where(st < 0)[0][:6]
agree = 0
ost = []
for k in range(0, count, 4):
    ro, xo = O.optimize(T.oracle_f(M.MODEL_EXP3_AFFINE, t, data[k]), 512, x0[k], dtype=np.float32)
    ost.append(ro.status)
print("oracle statuses on every 4th problem:", collections.Counter(ost))
for k in bad:
    ro, xo = O.optimize(T.oracle_f(M.MODEL_EXP3_AFFINE, t, data[k]), 512, x0[k], dtype=np.float32)
    print(k, "gpu", res[k], "oracle", O.STATUS[ro.status], ro.iterations, ro.fCalls, ro.residual)
