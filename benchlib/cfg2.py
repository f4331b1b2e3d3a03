"""bench.py --config cfg2: BASELINE cfg 2, the Gaussian-sum fit m = 1e5 x n = 16 in fp64 (resident-J path and launch
chain)."""
import ctypes as C
import json
import os
import sys
import time

from .common import (F64_MFMA_PEAK_TF, HBM_PEAK_GBS, ROOT, _round_no, csrc_sha16, describe_comm, flush_c_stdio,
                     # noqa: F401
                     pmc_field, pmc_file, step_stats, traffic_source)


def main_cfg2(args):
    """BASELINE cfg 2: Gaussian-sum curve fit, m = 1e5 residuals x n = 16 parameters, fp64, width bounds, FD Jacobian
    through the device callbacks (--fd batched: one launch for the 2n points of a refresh; --fd serial: one per point)
    (SURVEY 8d). J is 12.8 MB: every kernel of a pass is a few microseconds, so the solve is bound by launch latency and
    host round trips, not by HBM or MFMA -- the line reports the time per pass and per launch."""
    import numpy as np
    import torch

    import mir_optim_amd as M
    from mir_optim_amd import api, workloads as W
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import problems as P

    if not torch.cuda.is_available() or M.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: mir_optim_amd has no CPU path")
    g = P.gauss_sum(100000, K=5)
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    ws = api.lib().mir_lsq_workspace_create(g["m"], g["n"], 8)
    fdb = {"batched": True, "rowmajor": "rowmajor", "pointmajor": "pointmajor",
           "serial": False}[args.fd]   # 2n FD points per launch, or one call per point
    for _ in range(max(1, args.warmup)):
        res, x = prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, variant=args.variant, batched=fdb)
    # the timed region carries NO kernel events: at ~6 event pairs per round and 41 rounds per solve they cost 0.9 ms of
    # a 3 ms solve (scripts/ab_bench.sh); the per-kernel split comes from a second, instrumented pass of the same solves
    iters = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, x = prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, variant=args.variant, batched=fdb)
        iters += res.iterations
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = M.Stats()
    for _ in range(args.steps):
        prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, stats=st, variant=args.variant, flags=M.TIME_KERNELS,
                   batched=fdb)
    d = st.as_dict()
    K = args.steps
    m, n = g["m"], g["n"]
    rounds = d["solve_launches"] / K
    out = {
        "metric": "LM iterations/sec", "value": iters / dt, "unit": "iterations/s", "n_gpus": 1, "steps": K,
            "warmup": args.warmup,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
        "config": {"workload": f"cfg2 Gaussian-sum fit m={m} x n={n} fp64, width bounds, FD Jacobian ("
                               + ("batched device callback: the 2n points of a refresh in one launch" if fdb
                                  else "single-point device callback")
                               + "), default settings, whole solves",
                   "iterations_per_solve": iters / K, "passes_per_solve": d["passes"] / K, "rounds_per_solve": rounds,
                   "fcalls_per_solve": res.fCalls, "status": res.status.name, "residual": res.residual,
                   "qp_active_set_passes_per_solve": d["qp_active_set_passes"] / K,
                   "us_per_round": dt / K / max(1.0, rounds) * 1e6,
                   "time_split_ms_per_solve": {"caller_fd_callbacks": d["fd_callback_ms"] / K,
                                               "caller_trial_callbacks": d["trial_callback_ms"] / K,
                                               "jtj_kernels": d["jtj_ms"] / K, "solve_kernel": d["solve_ms"] / K,
                                               "total_wall_instrumented_pass": d["total_ms"] / K,
                                               "note": "from a second, event-instrumented pass (the timed region has "
                                                       "no events)"},
                   "parallelism": "replicas only at N > 1 (the problem is too small to shard)"},
        "roofline": {"kernel": "mirlsq::k_lm_solve<double, 1, true> (the n = 16 damped BOXCQP solve; the longest "
                               "library kernel of a round)",
                     "bound": "latency", "achieved": None, "peak": None, "unit": "us", "frac": None,
                     "avg_launch_ms": d["solve_ms"] / max(1, d["solve_launches"]), "launches": d["solve_launches"],
                         "traffic": None,
                     "note": "launch-latency bound: J^T J at n = 16 is 2 flop/B (SURVEY 8d) and J is 12.8 MB -- every "
                             "kernel of a "
                             "round runs for microseconds; the figure of merit is us_per_round"},
    }
    # ---- the resident-J path (include/mir_optim_amd_resident.hpp): the whole loop in ONE cooperative launch, J in the
    # CUs' LDS. It is the product path for a problem of this size; the launch chain timed above stays on the line as
    # `launch_chain`.
    chain = {k: out[k] for k in ("value", "ms_per_step")}
    chain.update({k: out["config"][k] for k in ("iterations_per_solve", "passes_per_solve", "rounds_per_solve",
                                                "us_per_round", "status",
                                                 "residual", "time_split_ms_per_solve")})
    chain["us_per_pass"] = dt / K / max(1.0, d["passes"] / K) * 1e6
    chain["solve_kernel"] = out.pop("roofline")
    rp = W.Resident.gauss_sum(g["t"], g["data"], K=5)
    if rp.plan_rc == 0:
        for _ in range(max(1, args.warmup)):
            rres, rx, rst = rp.solve(g["x0"], g["lower"], g["upper"])
        torch.cuda.synchronize()
        riters, steps_ms = 0, []
        t0 = time.perf_counter()
        for _ in range(K):
            ts = time.perf_counter()
            rp.upload_point(g["x0"], g["lower"],
                            g["upper"])       # x, lower, upper: 384 bytes, as the launch chain uploads them per solve
            # the timed solves carry no clock reads; the stamps come from the untimed one below
            rp.launch(variant=W.RESIDENT_NO_STAMPS)
            rp.stream.synchronize()
            steps_ms.append((time.perf_counter() - ts) * 1e3)
        rdt = time.perf_counter() - t0
        rres, rx, rst = rp.solve(g["x0"], g["lower"],
                                 g["upper"])  # the same solve once more for its result and in-kernel stamps
        riters = rres.iterations * K
        tick = 1e-2                                               # stats are in 10 ns ticks -> us
        rounds_r, passes_r = rst["rounds"], rst["passes"]
        # `value` is the resident path's (the product path for a problem of this size; it needs a compile-time residual
        # model); the launch chain's -- the one reachable through the reference's callback ABI -- stands beside it under
        # its own key
        out.update({"value": riters / rdt, "ms_per_step": rdt / K * 1e3, "value_path": "resident",
                    "value_resident": riters / rdt, "value_launch_chain": chain["value"],
                    "ms_per_step_resident": rdt / K * 1e3, "ms_per_step_launch_chain": chain["ms_per_step"]})
        out["config"].update({
            "workload": f"cfg2 Gaussian-sum fit m={m} x n={n} fp64, width bounds, FD Jacobian; resident-J path: the "
                        f"whole LM loop in one "
                        "cooperative launch, J / y / row data in the CUs' LDS, compile-time residual model, default "
                        "settings, whole solves",
            "path": "resident", "iterations_per_solve": rres.iterations, "passes_per_solve": passes_r,
                "rounds_per_solve": rounds_r,
            "fcalls_per_solve": rres.fCalls, "status": rres.status.name, "residual": rres.residual,
            "qp_active_set_passes_per_solve": rst["qp_active_set_passes"],
            "us_per_round": rdt / K / max(1, rounds_r) * 1e6, "us_per_pass": rdt / K / max(1, passes_r) * 1e6,
            "step_ms_min_median_max": [float(np.min(steps_ms)), float(np.median(steps_ms)), float(np.max(steps_ms))],
            "grid": rst["grid"], "rows_per_workgroup": rst["rows"], "lds_bytes_per_workgroup": rp.lds_bytes,
            "kernel_us_per_solve": rst["t_total"] * tick,
            "time_split_us_per_solve": {"workers_trial_residuals": rst["t_w_eval"] * tick,
                                        "workers_fd_refreshes": rst["t_w_fd"] * tick,
                                        "workers_products_and_publication": rst["t_w_prod"] * tick,
                                        "group_leaders": rst["t_group"] * tick,
                                            "wait_for_totals": rst["t_total_wait"] * tick,
                                        "solver_workgroup": rst["t_solver"] * tick,
                                            "of_which_n_x_n_solves": rst["t_solve_body"] * tick,
                                        "wait_for_command": rst["t_cmd_wait"] * tick, "staging": rst["t_stage"] * tick,
                                        "note": "stamps of workgroup 0 (s_memrealtime) inside the one launch"},
            "jacobian_full": rst["jacobian_full"], "jacobian_broyden": rst["jacobian_broyden"],
                "rejected": rst["rejected"],
            "elided_null_steps": rst["elided_evaluations"],
                "rejections_decided_by_lookahead": rst["lookahead_rejections"]})
        out["config"].pop("time_split_ms_per_solve", None)
        out["launch_chain"] = chain
        # HBM bytes of one launch from the committed rocprofv3 --pmc passes of this command (FETCH_SIZE doubled: the
        # gfx950 correction)
        traffic, traffic_src, valu_busy = None, None, None
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cfg2_pmc.json")), key=_round_no):
            try:
                pm = next(v for k, v in json.load(open(f))["kernels"].items() if "k_lm_resident" in k)
                traffic, traffic_src, valu_busy = pm.get("hbm_bytes_per_launch"), os.path.relpath(f,
                        ROOT), pm.get("valu_util")
            except (StopIteration, KeyError, ValueError):
                pass
        out["roofline"] = {"kernel": "mirlsq::k_lm_resident<ResGaussSum<5>, true> (the one launch of a solve)",
                           "bound": "latency", "achieved": None,
                           "peak": None, "unit": "us", "frac": None, "avg_launch_ms": rst["t_total"] * tick / 1e3,
                               "launches": 1, "traffic": traffic,
                           "traffic_source": traffic_src, "valu_busy_pmc": valu_busy,
                           "algorithmic_bytes_per_launch": float(m * 2 * 8 + 3 * n * 8),
                           "note": "J never leaves LDS: 16 MB of operands against 40 MB of LDS on the chip; a pass is "
                                   "three in-launch hand-offs "
                                   "(members -> 16 leaders -> workgroup 0 -> everybody) and a one-wave n = 16 solve -- "
                                   "latency, not HBM or MFMA. "
                                   "Figures of merit: us_per_pass, us_per_round"}
        res, x = rres, rx
    else:
        out["roofline"] = chain["solve_kernel"]
        out["value_path"] = "launch chain"
        out["value_launch_chain"] = chain["value"]
        out["config"]["path"] = f"launch chain (resident plan returned {rp.plan_rc})"
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        ctx = O.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
        t1 = time.perf_counter()
        ro, xo = O.optimize(O.native_fn("wlc_gauss_sum_f"), m, g["x0"], lower=g["lower"], upper=g["upper"],
                            fctx=C.addressof(ctx))
        dtc = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": ro.iterations / dtc, "unit": "iterations/s",
                               "cores": int(os.environ.get("OMP_NUM_THREADS", "1")),
                               "host_nproc": os.cpu_count(), "kind": "port",
                               "sample": f"the whole solve ({ro.iterations} iterations, fCalls {ro.fCalls}, status "
                                         f"{O.STATUS.get(ro.status)}), {dtc:.2f} s, "
                                         "plain-loop BLAS, OpenMP residuals",
                               "parity_x_max_abs_diff": float(np.abs(np.asarray(x) - np.asarray(xo)).max()),
                               "parity_residual_rel_diff": abs(res.residual - ro.residual) / abs(ro.residual)}
    api.lib().mir_lsq_workspace_destroy(ws)
    print(json.dumps(out), flush=True)
