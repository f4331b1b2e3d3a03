"""Runs the CPU oracle on the full-size cfg-3 problem in its OWN process (tests/test_gpu_fullsize.py): the oracle's OpenMP
regions and OpenBLAS's thread pool then live in a fresh process -- the set-up bench.py's cpu_baseline leg uses -- instead of
sharing one with the GPU runtime and everything the earlier tests of a pytest session have loaded.
usage: oracle_fullsize_worker.py m n abs_tolerance max_iterations threads out.npz"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    m, n, abs_tol, max_it, threads, out = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import faulthandler
    faulthandler.enable()
    import numpy as np
    from mir_optim_amd import workloads as W
    from oracle import oracle as O
    data = W.tanh_linear_data(m, n)
    ob = O.load_openblas(threads=threads)
    so = O.default_settings()
    so.absTolerance = abs_tol
    so.maxIterations = max_it
    ctx = O.TanhLinearCtx(data["A"].ctypes.data, data["b"].ctypes.data)
    events = []                                  # the oracle's per-pass trace: (event, iterations, lambda, residual, trial, dx.dx)
    t0 = time.perf_counter()
    ro, xo = O.optimize(O.native_fn("wlc_tanh_linear_f"), m, data["x0"], settings=so, fctx=C.addressof(ctx), use_openblas=ob,
                        trace=lambda *a: events.append(a))
    dt = time.perf_counter() - t0
    np.savez(out, x=xo, status=ro.status, iterations=ro.iterations, fCalls=ro.fCalls, residual=ro.residual, seconds=dt,
             openblas=int(bool(ob)), threads=threads, trace=np.array(events, dtype=np.float64).reshape(-1, 6))


if __name__ == "__main__":
    main()
