"""Time the caller-side batched residual GEMM alone (m = 1e6, n = 128, 256 points, row-major panel): HIP events, 20 launches."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mir_optim_amd as M
from mir_optim_amd import api, workloads as W
m, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128
d = W.tanh_linear_data(m, n)
prob = W.TanhLinear(d["A"], d["b"])
p = 2 * n
X = np.tile(d["x0"], (p, 1)); X[np.arange(p), np.arange(p) // 2] += 1e-8 * (1 - 2 * (np.arange(p) % 2))
dX = api.DeviceBuffer(X); dY = api.DeviceBuffer(nbytes=m * p * 8, dtype=np.float64, shape=(m, p))
WL = api.workloads_lib()
ctx = C.c_void_p(C.addressof(prob.ctx))
s = torch.cuda.ExternalStream(prob.stream.handle)
def call():
    WL.wl_tanh_linear_fbr_d(ctx, C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr), C.c_void_p(dY.ptr))
for _ in range(3): call()
prob.stream.synchronize()
with torch.cuda.stream(s):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(20): call()
    b.record(s)
prob.stream.synchronize()
ms = a.elapsed_time(b) / 20
print(f"m={m} gemm {ms:.4f} ms  {2.0 * m * n * p / ms / 1e9:.1f} TF")
