"""Randomised sweep of the wave-per-problem kernel (cfg 5's model, n = 8, fp32) against the FUSED float oracle, bit for bit, on
shapes the tiers do not pin: row counts 1 .. 1400 (ragged against the 64 lanes and the chunks of 8 loads), 1 .. 48 problems a
launch, noise levels from 0 to 0.1, starts near and far, maxIterations from 1 up.  python scripts/fuzz_batched.py [cases=200] [seed0=0]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import mir_optim_amd as M
from mir_optim_amd import api
from oracle import oracle as O


def run(cases, seed0, verbose=True):
    """-> (fits compared, fits that are not bit-identical)"""
    L = api.lib()
    n = 8
    bad = 0
    fits = 0
    for k in range(cases):
        rng = np.random.default_rng(seed0 + k)
        m = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 127, 128, 129, 511, 512, 513, int(rng.integers(1, 1400))]))
        count = int(rng.integers(1, 49))
        t = np.sort(rng.random(m) * float(rng.choice([1.0, 4.0, 10.0]))).astype(np.float32)
        truth = np.column_stack([1 + rng.random(count), 0.2 + 2 * rng.random(count), rng.random(count) - 0.5] + [0.3 * (rng.random(count) - 0.5) for _ in range(5)])
        tt = t.astype(np.float64)
        clean = (truth[:, 0:1] * np.exp(-tt[None, :] * truth[:, 1:2]) + truth[:, 2:3] + truth[:, 3:4] * np.sin(2 * tt) + truth[:, 4:5] * np.cos(2 * tt)
                 + truth[:, 5:6] * np.sin(5 * tt) + truth[:, 6:7] * np.cos(5 * tt) + truth[:, 7:8] * tt)
        data = (clean + float(rng.choice([0.0, 1e-4, 1e-2, 0.1])) * (2 * rng.random((count, m)) - 1)).astype(np.float32)
        x0 = (truth * (1 + float(rng.choice([0.01, 0.1, 0.5])) * (2 * rng.random((count, n)) - 1))).astype(np.float32)
        s = M.LeastSquaresSettings(np.float32)
        s.maxIterations = int(rng.choice([1, 3, 20, 1000]))
        s.maxAge = int(rng.choice([0, 0, 1, 4]))
        dt_, dd, dx = api.DeviceBuffer(t), api.DeviceBuffer(data), api.DeviceBuffer(x0)
        dlo = api.DeviceBuffer(np.full(n, -np.inf, dtype=np.float32)); dup = api.DeviceBuffer(np.full(n, np.inf, dtype=np.float32))
        dres = api.DeviceBuffer(nbytes=count * 24, dtype=np.uint8, shape=(count * 24,))
        dbasis = api.DeviceBuffer(nbytes=m * 16, dtype=np.float32, shape=(m, 4))
        st = api.Stream()
        opt = api.BatchedOptions(stream=st.handle, basis=dbasis.ptr, basis_bytes=m * 16)
        rc = L.mir_lsq_batched_kernel_s(C.byref(s), count, m, M.MODEL_EXP_DECAY_PAD8, dx.ptr, dlo.ptr, dup.ptr, dt_.ptr, 0, dd.ptr, dres.ptr, C.byref(opt))
        assert rc == 0, rc
        st.synchronize()
        raw = np.frombuffer(dres.download().tobytes(), dtype=np.dtype([("status", "<i4"), ("iterations", "<u4"), ("fCalls", "<u4"), ("gCalls", "<u4"),
                                                                       ("residual", "<u4"), ("lambda", "<u4")])).copy()
        x = dx.download().reshape(count, n).copy()
        basis = dbasis.download().reshape(m, 4).copy()
        for b in (dt_, dd, dx, dlo, dup, dres, dbasis):
            b.free()
        for p in range(count):
            (so, ito, fco, gco, ro, lo_), xo = O.optimize_batched_fused_pad8_s(s, t, basis, data[p], x0[p])
            fits += 1
            same = (so == raw["status"][p] and ito == raw["iterations"][p] and fco == raw["fCalls"][p]
                    and int(np.float32(ro).view(np.uint32)) == int(raw["residual"][p]) and int(np.float32(lo_).view(np.uint32)) == int(raw["lambda"][p])
                    and xo.tobytes() == x[p].tobytes())
            if not same:
                bad += 1
                if verbose and bad <= 20:
                    print(f"MISMATCH seed {seed0 + k} m {m} count {count} problem {p} maxIt {s.maxIterations} maxAge {s.maxAge}: gpu ({int(raw['status'][p])}, {int(raw['iterations'][p])}, "
                          f"{int(raw['fCalls'][p])})  oracle ({so}, {ito}, {fco})  |dx| {np.abs(xo - x[p]).max():.3e}", flush=True)
    return fits, bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    fits, bad = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"summary: {fits} fits in {cases} launches, {bad} not bit-identical")
