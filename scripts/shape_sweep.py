"""Rooflines of the shapes outside the two tuned ones: the same tanh-linear problem family solved in
f32 (general solver), f64 with an odd n, and n = 512, with the event-timed per-kernel split of mir_lsq_stats turned into
achieved GB/s / TFLOP/s against the MI355X peaks (HBM 8 TB/s; dense MFMA f64 78.6 TF, f32 157.3 TF).
One JSON line per shape. usage (GPU box): python scripts/shape_sweep.py > gpurun_out/shape_sweep.jsonl"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import workloads as W

HBM, MFMA = 8000.0, {8: 78.6, 4: 157.3}
shapes = [(1_000_000, 128, np.float64, "tuned reference shape"), (1_000_000, 128, np.float32, "f32 through the general solver"),
          (1_000_000, 127, np.float64, "odd n"), (1_000_000, 96, np.float64, "n % 32 != 0"), (250_000, 512, np.float64, "n > 256"),
          (1_000_000, 256, np.float64, "cfg 4's per-GPU shape"), (1_000_000, 64, np.float32, "f32, n = 64"),
          (100_000, 1024, np.float64, "n = 1024: above the read-only Broyden sweep, everything through the any-n kernels")]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if str(s[1]) in sys.argv[1:] or s[2].__name__ in sys.argv[1:]]
for m, n, dt, note in shapes:
    T = np.dtype(dt).itemsize
    d = W.tanh_linear_data(m, n)
    prob = W.TanhLinear(d["A"].astype(dt), d["b"].astype(dt), dtype=dt)
    s = M.LeastSquaresSettings(dt); s.absTolerance = 1e-5 if T == 8 else 1e-3
    best = None
    for rep in range(3):
        st = M.Stats()
        t0 = time.perf_counter()
        res, x = prob.solve(d["x0"].astype(dt), settings=s, batched=True, stats=st, flags=M.TIME_KERNELS)
        dtm = (time.perf_counter() - t0) * 1e3
        if best is None or dtm < best[0]:
            best = (dtm, res, st.as_dict())
    dtm, res, q = best
    nfd, nbr = q["jtj_fd_launches"], q["jtj_broyden_launches"]
    npl = q["jtj_launches"] - nfd - nbr
    out = {"m": m, "n": n, "dtype": dt.__name__, "note": note, "solve_ms": dtm, "status": res.status.name, "iterations": int(res.iterations),
           "fcalls": int(res.fCalls), "refreshes": int(q["jacobian_full"]), "broyden_passes": int(q["jacobian_broyden"])}
    jflop = m * n * (n + 1.0) + 2.0 * m * n
    if nfd:
        ms = q["jtj_fd_ms"] / nfd
        by = T * (2.0 * m * n + m) if q["fd_callback_points"] else 0
        out["fd_jtj_kernel"] = {"avg_ms": ms, "launches": int(nfd), "TFLOPs": jflop / ms / 1e9, "mfma_frac": jflop / ms / 1e9 / MFMA[T],
                                "GBs_min": T * (2.0 * m * n + m) / ms / 1e6, "hbm_frac_min": T * (2.0 * m * n + m) / ms / 1e6 / HBM,
                                "note": "bytes: the m x n difference panel read + J written (the pair panel reads twice that)"}
    if npl:
        ms = (q["jtj_ms"] - q["jtj_fd_ms"] - q["jtj_broyden_ms"]) / npl
        out["plain_jtj_kernel"] = {"avg_ms": ms, "launches": int(npl), "TFLOPs": jflop / ms / 1e9, "mfma_frac": jflop / ms / 1e9 / MFMA[T],
                                   "GBs": T * (m * n + m) / ms / 1e6, "hbm_frac": T * (m * n + m) / ms / 1e6 / HBM}
    if nbr:
        ms = q["jtj_broyden_ms"] / nbr
        kbar = q["broyden_lr_columns"] / nbr
        lowrank = n <= 512            # kLrMaxN (csrc/common.h)
        by = T * (m * n + (kbar + 3) * m) if lowrank else T * (2.0 * m * n + 3 * m)
        out["broyden_kernel"] = {"avg_ms": ms, "launches": int(nbr), "GBs": by / ms / 1e6, "hbm_frac": by / ms / 1e6 / HBM,
                                 "kind": "read-only sweep (k_broyden_lr)" if lowrank else "rewrite + tile-pair J^T J (n > 512)"}
    out["solve_kernel"] = {"avg_ms": q["solve_ms"] / max(1, q["solve_launches"]), "launches": int(q["solve_launches"])}
    if q["fd_callback_calls"]:
        out["caller_fd"] = {"ms_per_refresh": q["fd_callback_ms"] / max(1, q["jacobian_full"]), "calls_per_refresh": q["fd_callback_calls"] / max(1, q["jacobian_full"])}
    if q["trial_callback_calls"]:
        ms = q["trial_callback_ms"] / q["trial_callback_calls"]
        out["caller_trial"] = {"avg_ms": ms, "GBs": T * (m * n + 2.0 * m) / ms / 1e6}
    print(json.dumps(out), flush=True)
    prob.dA.free(); prob.db.free()
