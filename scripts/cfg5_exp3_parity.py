"""cfg 5, the ill-conditioned stress model EXP3_AFFINE (three exponentials + affine, fp32): the wave kernel against the float
oracle on ALL 4096 problems -- the distribution the test tolerances of tests/test_gpu_batched.py are taken from.
usage (GPU box): python scripts/cfg5_exp3_parity.py [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mir_optim_amd as M
from oracle import oracle as O
import test_gpu_batched as TB

count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
O.build()
t, data, truth, x0 = TB.make_exp3(count)
res, x = M.optimizeLeastSquaresBatched(M.MODEL_EXP3_AFFINE, x0, t, data)
st = np.array([int(r.status) for r in res]); rg = np.array([r.residual for r in res])
so = np.empty(count, dtype=int); ro_ = np.empty(count); xo = np.empty_like(x)
for k in range(count):
    r, xk = O.optimize(TB.oracle_f(M.MODEL_EXP3_AFFINE, t, data[k]), 512, x0[k], dtype=np.float32)
    so[k], ro_[k], xo[k] = r.status, r.residual, xk
both = (st >= 0) & (so >= 0)
print("problems", count, " gpu ok", int((st >= 0).sum()), " oracle ok", int((so >= 0).sum()), " both ok", int(both.sum()))
ratio = rg[both] / ro_[both]
print("residual ratio gpu/oracle (both ok): min %.4f  q01 %.4f  median %.6f  q99 %.4f  max %.4f" % (ratio.min(), np.quantile(ratio, .01), np.median(ratio), np.quantile(ratio, .99), ratio.max()))
td = t.astype(np.float64)
def model(p):
    return p[:, 0:1] * np.exp(-td * p[:, 1:2]) + p[:, 2:3] * np.exp(-td * p[:, 3:4]) + p[:, 4:5] * np.exp(-td * p[:, 5:6]) + p[:, 6:7] + p[:, 7:8] * td
cd = np.abs(model(x[both].astype(np.float64)) - model(xo[both].astype(np.float64))).max(axis=1)
print("max |fitted curve gpu - oracle| (both ok): median %.3e  q99 %.3e  max %.3e   (noise amplitude 2e-3)" % (np.median(cd), np.quantile(cd, .99), cd.max()))
xe = (np.abs(x[both] - xo[both]) / np.maximum(1, np.abs(xo[both]))).max(axis=1)
print("parameter difference rel: median %.3e  q90 %.3e  q99 %.3e  max %.3e" % (np.median(xe), np.quantile(xe, .9), np.quantile(xe, .99), xe.max()))
only_g = (st >= 0) & (so < 0); only_o = (st < 0) & (so >= 0)
print("gpu ok / oracle failed:", int(only_g.sum()), "  oracle ok / gpu failed:", int(only_o.sum()))
