"""Where does a cfg 5 fit spend its cycles? Needs a profiling build of the solver library:
   MIR_OPTIM_AMD_CXXFLAGS=-DMIRLSQ_BATCHED_TIMING hipcc ... -o tmp_ab/libT.so   (see batched_kernel.h, MIRLSQ_BATCHED_TIMING)
usage: python scripts/probes/cfg5_phases.py tmp_ab/libT.so"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import mir_optim_amd.build as B
B.SOLVER_LIB = os.path.abspath(sys.argv[1])
B.build = lambda *a, **k: (B.SOLVER_LIB, B.WORKLOADS_LIB)
import mir_optim_amd as M
from mir_optim_amd import api
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P

count, m, n = 4096, 512, 8
t, data, truth, x0 = P.cfg5_pad8(count, m)
L = api.lib()
s = M.LeastSquaresSettings(np.float32)
dt_, dd, dx = api.DeviceBuffer(t), api.DeviceBuffer(data), api.DeviceBuffer(x0)
dlo = api.DeviceBuffer(np.full(n, -np.inf, dtype=np.float32)); dup = api.DeviceBuffer(np.full(n, np.inf, dtype=np.float32))
dres = api.DeviceBuffer(nbytes=count * 24, dtype=np.uint8, shape=(count * 24,))
st = api.Stream()
dtm = api.DeviceBuffer(np.zeros((count, 10), dtype=np.uint64))      # mir_lsq_batched_options.timing: 10 counters per problem
opt = api.BatchedOptions(stream=st.handle, timing=dtm.ptr)
for rep in range(2):
    dx.upload(x0)
    assert L.mir_lsq_batched_kernel_s(C.byref(s), count, m, M.MODEL_EXP_DECAY_PAD8, dx.ptr, dlo.ptr, dup.ptr, dt_.ptr, 0, dd.ptr, dres.ptr, C.byref(opt)) == 0
    st.synchronize()
tm = dtm.download().reshape(count, 10).astype(np.float64)
names = ["residual evaluations", "Jacobian refresh (FD / Broyden)", "J^T J, J^T y + reductions", "damped solves", "whole fit"]
tot = tm[:, 4]
print("s_memtime ticks (shader clock); mean per fit, share of the fit")
for k in range(4):
    print(f"  {names[k]:34s} {tm[:, k].mean():10.0f}  {tm[:, k].sum() / tot.sum():6.1%}")
for k, nm in ((6, "a trial's preparation"), (7, "an accepted step's bookkeeping")):
    print(f"  {nm:34s} {tm[:, k].mean():10.0f}  {tm[:, k].sum() / tot.sum():6.1%}   (part of everything else)")
print(f"  {'everything else':34s} {(tot - tm[:, :4].sum(axis=1)).mean():10.0f}  {1 - tm[:, :4].sum() / tot.sum():6.1%}")
print(f"  whole fit: mean {tot.mean():.0f}, max {tot.max():.0f} ticks; solve calls per fit {tm[:, 5].mean():.2f}")
i = int(np.argmax(tot))
print("  the longest fit:", {names[k]: int(tm[i, k]) for k in range(5)}, "solves", int(tm[i, 5]))
raw = np.frombuffer(dres.download().tobytes(), dtype=np.dtype([("status", "<i4"), ("iterations", "<u4"), ("fCalls", "<u4"), ("gCalls", "<u4"),
                                                               ("residual", "<f4"), ("lambda", "<f4")]))
print("  iterations per fit", raw["iterations"].mean(), "fCalls per fit", raw["fCalls"].mean())
