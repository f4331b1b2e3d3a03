"""cfg 5: where does the launch's time go? Times the 4096-fit launch as it is, the same problems in other ORDERS (longest
fits first / last), subsets (the 2048 that fit the slots at once; the longest fit alone), and prints the distribution of the
fits' lengths. usage: python scripts/cfg5_tail.py"""
import ctypes as C
import os
import sys
import time

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."),
                os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import numpy as np

import mir_optim_amd as M
from mir_optim_amd import api
import problems as P

count, m, n = 4096, 512, 8
t, data, truth, x0 = P.cfg5_pad8(count, m)
L = api.lib()
s = M.LeastSquaresSettings(np.float32)
stream = api.Stream()
basis = api.DeviceBuffer(nbytes=m * 16, dtype=np.uint8, shape=(m * 16,))
bopt = api.BatchedOptions(stream=stream.handle, basis=basis.ptr, basis_bytes=m * 16)
dt_ = api.DeviceBuffer(t)
dlo = api.DeviceBuffer(np.full(n, -np.inf, dtype=np.float32))
dup = api.DeviceBuffer(np.full(n, np.inf, dtype=np.float32))
rdt = np.dtype([("status", "<i4"), ("iterations", "<u4"), ("fCalls", "<u4"), ("gCalls", "<u4"), ("residual", "<f4"), ("lambda", "<f4")])


def run(order, reps=40, label=""):
    cnt = len(order)
    d = np.ascontiguousarray(data[order]); x = np.ascontiguousarray(x0[order])
    dd, dx0, dx = api.DeviceBuffer(d), api.DeviceBuffer(x), api.DeviceBuffer(x)
    dres = api.DeviceBuffer(nbytes=cnt * 24, dtype=np.uint8, shape=(cnt * 24,))

    def step():
        L.mir_lsq_memcpy_d2d(dx.ptr, dx0.ptr, cnt * n * 4, stream.handle)
        rc = L.mir_lsq_batched_kernel_s(C.byref(s), cnt, m, M.MODEL_EXP_DECAY_PAD8, dx.ptr, dlo.ptr, dup.ptr, dt_.ptr, 0,
                                        dd.ptr, dres.ptr, C.byref(bopt))
        assert rc == 0, rc
    for _ in range(3):
        step()
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    stream.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    raw = np.frombuffer(dres.download().tobytes(), dtype=rdt)
    print(f"{label:44s} {cnt:5d} fits  {ms:7.3f} ms   iterations sum {int(raw['iterations'].sum())}  fCalls sum {int(raw['fCalls'].sum())}", flush=True)
    for b in (dd, dx0, dx, dres):
        b.free()
    return raw


raw = run(np.arange(count), label="as it is (index order)")
work = raw["fCalls"].astype(np.int64)          # a fit's length: residual evaluations (FD refreshes count n)
it = raw["iterations"]
print("iterations: mean %.1f  median %d  p90 %d  p99 %d  max %d;  fCalls: mean %.1f p99 %d max %d" % (
    it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max(), work.mean(), np.percentile(work, 99), work.max()))
desc = np.argsort(-work, kind="stable")
run(desc, label="longest fits first")
run(desc[::-1], label="longest fits last")
run(np.arange(2048), label="first 2048 (all resident at once)")
run(desc[:1], reps=200, label="the longest fit alone")
run(desc[:256], label="the 256 longest, one per CU")
run(desc[:1024], label="the 1024 longest, one per SIMD")
run(desc[:2048], label="the 2048 longest, two per SIMD")
med = np.argsort(np.abs(work - np.median(work)))[:2048]
run(med, label="2048 fits of median length")
