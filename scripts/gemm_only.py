"""Time the caller-side batched residual GEMM alone (2n finite-difference points, the m x n difference panel of
fbRowMajorDiff): HIP events, 20 launches.  usage: gemm_only.py [m] [n] [path of another build of the workload library]
(the third argument is for same-box A/B runs of two builds)."""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mir_optim_amd as M
from mir_optim_amd import api, workloads as W
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = W.tanh_linear_data(m, n)
prob = W.TanhLinear(d["A"], d["b"])
p = 2 * n
X = np.tile(d["x0"], (p, 1)); X[np.arange(p), np.arange(p) // 2] += 1e-8 * (1 - 2 * (np.arange(p) % 2))
dX = api.DeviceBuffer(X); dY = api.DeviceBuffer(nbytes=m * n * 8, dtype=np.float64, shape=(m, n))
# RTLD_DEEPBIND: the in-tree library is loaded RTLD_GLOBAL, and without it the other build's calls across its own translation
# units (wl_* -> launch_*) would bind to the in-tree definitions -- an A/B of one build against itself
WL = C.CDLL(sys.argv[3], mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND) if len(sys.argv) > 3 else api.workloads_lib()
ctx = C.c_void_p(C.addressof(prob.ctx))
s = torch.cuda.ExternalStream(prob.stream.handle)
def call():
    WL.wl_tanh_linear_fbd_d(ctx, C.c_size_t(m), C.c_size_t(n), C.c_size_t(p), C.c_void_p(dX.ptr), C.c_void_p(dY.ptr))
for _ in range(3): call()
prob.stream.synchronize()
with torch.cuda.stream(s):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(20): call()
    b.record(s)
prob.stream.synchronize()
ms = a.elapsed_time(b) / 20
print(f"m={m} n={n} {os.path.basename(sys.argv[3]) if len(sys.argv) > 3 else 'in-tree build'}: gemm {ms:.4f} ms  {2.0 * m * n * p / ms / 1e9:.1f} TF")
