import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import mir_optim_amd as M
from mir_optim_amd import workloads as W
m, n = 1_000_000, 128
d = W.tanh_linear_data(m, n)
for dt in (np.float64, np.float32):
    prob = W.TanhLinear(d["A"].astype(dt), d["b"].astype(dt), dtype=dt)
    s = M.LeastSquaresSettings(dt); s.absTolerance = 1e-5 if dt == np.float64 else 1e-3
    for rep in range(3):
        st = M.Stats()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, x = prob.solve(d["x0"].astype(dt), settings=s, batched=True, stats=st, flags=M.TIME_KERNELS)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    dd = st.as_dict()
    print(dt.__name__, res, f"{(t1-t0)*1e3:.2f} ms", {k: round(v, 3) for k, v in dd.items() if k.endswith("_ms") and v})
