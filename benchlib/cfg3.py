"""bench.py default: BASELINE cfg 3, the tall-skinny synthetic NLS m = 1e6 x n = 128 in fp64 -- whole solves through the
device-callback contract, HIP-event timing of the hot kernels, the CPU baseline legs, N ranks over RCCL."""
import ctypes as C
import json
import os
import sys
import time

from .common import (F64_MFMA_PEAK_TF, HBM_PEAK_GBS, ROOT, _round_no, csrc_sha16, describe_comm, flush_c_stdio,
                     # noqa: F401
                     pmc_field, pmc_file, step_stats, traffic_source)


def main_cfg3(args):
    """BASELINE cfg 3 (and cfg 4's per-GPU shape with --n 256 / --scaling weak): one rank of the benchmark."""
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))   # CPU baseline leg (oracle, OpenMP)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` "
                         f"(self-launching) or "
                         "torch.distributed.run --nproc-per-node N")
    import numpy as np
    import torch
    import torch.distributed as dist

    import mir_optim_amd as M
    from mir_optim_amd import api, parallel as PAR, workloads as W

    if not torch.cuda.is_available() or M.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: mir_optim_amd has no CPU path")
    # rehearsals: ranks share the GPUs there are (asked for, or forced: fewer visible devices than ranks -- RCCL then
    # refuses the communicator and the run takes the labelled callback fallback below instead of dying in set_device)
    ndev = torch.cuda.device_count()
    share = args.comm == "gloo-callback" or os.environ.get("BENCH_SHARE_GPU") == "1" or ndev < world
    if ndev < world and rank == 0:
        print(f"[bench] {world} ranks on {ndev} visible GPU(s): ranks share devices (rehearsal, not a scaling "
              f"measurement)",
              file=sys.stderr, flush=True)
    torch.cuda.set_device(local_rank % ndev if share else local_rank)
    comm = None
    comm_obj = None
    comm_fallback = None
    t_comm = None
    distributed = world > 1 or args.force_comm or os.environ.get("MIR_LSQ_FORCE_COMM") == "1"
    ctl_dev = "cpu"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # Control plane (unique-id exchange, the barriers around the timed region, the max over ranks):
        # torch.distributed. Data plane (every collective of the solve): the solver's OWN RCCL communicator over xGMI,
        # created below.
        if args.control_plane == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            ctl_dev = "cuda"
        else:
            dist.init_process_group("gloo")

        def bcast(buf):
            t = torch.from_numpy(buf).to(ctl_dev)
            dist.broadcast(t, 0)
            return t.cpu().numpy()
        comm_fallback = None
        if args.comm == "rccl":
            # the solver's own RCCL communicator (xGMI), id via torch.distributed; checked with one all-reduce of a
            # known payload before anything is timed. If ANY rank fails to create or verify it, every rank falls back to
            # the callback communicator over the control plane -- slower, labelled in config, but a measured line
            # instead of a crash.
            err = None
            try:
                comm = PAR.rccl_comm(world, rank, bcast)
                if api.lib().mir_lsq_comm_ranks(comm) != world:
                    err = f"ncclCommCount = {api.lib().mir_lsq_comm_ranks(comm)}, expected {world}"
                elif not PAR.check_comm(comm, world, rank):
                    err = "all-reduce self-check returned wrong sums"
            except Exception as e:      # noqa: BLE001
                err = repr(e)
            flag = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() > 0:
                print(f"[bench] rank {rank}: RCCL communicator unusable ({err or 'on another rank'}); falling back to "
                      f"the callback "
                      "communicator over torch.distributed", file=sys.stderr, flush=True)
                if comm:
                    api.lib().mir_lsq_comm_destroy(comm)
                comm = None
                comm_fallback = err or "failure on another rank"
                args.comm = "gloo-callback"
            else:
                t_comm = time.perf_counter()
        if args.comm == "gloo-callback":
            comm_obj = PAR.HostAllreduceComm(world, rank, PAR.torch_allreduce_numpy(dist))
            comm = comm_obj.handle

    n = args.n
    replay = args.replay_ranks if (world == 1 and args.replay_ranks > 1) else 0
    if args.scaling == "strong":
        m_total = args.rows
        row0, m = PAR.row_shard(m_total, max(world, replay), rank)
    else:
        m = args.rows
        m_total = m * world
        row0 = rank * m
    data = W.tanh_linear_data(m, n, row_offset=row0)
    prob = W.TanhLinear(data["A"], data["b"])
    prob.ctx.read_a_once = 1 if args.gemm_read_a_once else 0
    settings = M.LeastSquaresSettings()
    settings.absTolerance = args.abs_tolerance
    ws = api.lib().mir_lsq_workspace_create(m, n, 8)
    if not ws:
        raise SystemExit("workspace allocation failed")
    fdb = {"batched": True, "rowmajor": "rowmajor", "pointmajor": "pointmajor", "serial": False}[args.fd]
    replay_info = None
    inner_comm = None
    if replay:
        if args.scaling != "strong":
            raise SystemExit("--replay-ranks measures a strong-scaled rank")

        def shard(r):
            o, ml = PAR.row_shard(m_total, replay, r)
            d = W.tanh_linear_data(ml, n, row_offset=o)
            return W.TanhLinear(d["A"], d["b"])
        tape, rres, rx, rwall = PAR.record_rank_tape(shard, replay, data["x0"], settings=settings, batched=fdb,
                                                     variant=args.variant)
        # --force-comm: the one-rank RCCL communicator created above
        inner = comm
        inner_comm = inner
        comm = PAR.replay_comm(replay, 0, tape, inner)
        if args.replay_latency_us and api.lib().mir_lsq_comm_replay_set_delay(comm, args.replay_latency_us) != 0:
            raise SystemExit("mir_lsq_comm_replay_set_delay failed")
        replay_info = {"ranks": replay, "tape_doubles": int(tape.size), "grouped_solve_wall_ms": rwall * 1e3,
                       "modelled_latency_us_per_exchange": args.replay_latency_us or None,
                       "grouped_solve": {"status": rres.status.name, "iterations": rres.iterations,
                                         "fcalls": rres.fCalls,
                                         "residual": rres.residual},
                       "inner": "one-rank RCCL all-reduce behind every replayed exchange" if inner else None,
                       "note": "value = iterations of the GLOBAL solve per second as ONE rank would deliver them with "
                               "a zero-latency "
                               "interconnect: rank 0's shard on an otherwise idle GPU, every all-reduce replaced by "
                               "the recorded total "
                               "of the real " + str(replay) + "-shard run (stream-ordered device copy)"}
        args.survey_steps = 0                                   # the tape belongs to the headline settings

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def solve(stats=None, flags=0, s=settings):
        if replay:
            api.lib().mir_lsq_comm_replay_rewind(comm)
        return prob.solve(data["x0"], settings=s, stats=stats, flags=flags, comm=comm_obj or comm, workspace=ws,
                          variant=args.variant,
                          batched=fdb)

    # RCCL finishes part of its initialisation asynchronously: a few seconds after ncclCommInitRank every HIP launch of
    # the process stalls once or twice for 60-150 ms (measured on a one-GPU box with --force-comm: 160-310 it/s when
    # that lands in the timed region, 740-760 when not; RCCL / NCCL knobs and warm collectives do not move it). Instead
    # of sleeping a fixed time, run untimed solves and WATCH for it: a solve that takes more than 4x the fastest one
    # seen is the stall; the loop ends once a stall has been seen and 20 solves in a row are back to normal, or at
    # --stall-bound seconds after communicator creation. All ranks take the same decision (the flag is max-reduced over
    # the control plane).
    stall = {"observed": 0, "max_ms": 0.0, "waited_s": 0.0, "solves": 0}
    if t_comm is not None and args.stall_bound > 0:
        best, calm = None, 0
        while True:
            t1 = time.perf_counter()
            solve()
            dt1 = time.perf_counter() - t1
            stall["solves"] += 1
            best = dt1 if best is None else min(best, dt1)
            if stall["solves"] > 2 and dt1 > 4 * best:
                stall["observed"] += 1
                stall["max_ms"] = max(stall["max_ms"], dt1 * 1e3)
                calm = 0
            else:
                calm += 1
            done = (stall["observed"] > 0 and calm >= 20) or (time.perf_counter() - t_comm) >= args.stall_bound
            flag = torch.tensor([1.0 if done else 0.0], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() > 0:
                break
        stall["waited_s"] = time.perf_counter() - t_comm
    for _ in range(args.warmup):
        res, x = solve()
    flush_c_stdio()     # every rank: RCCL's init banner leaves the C stdio buffer now, not at process exit

    def timed(count, s):
        # two statistics records: `st` for the steps whose kernels are bracketed with HIP events (its per-launch figures
        # -- milliseconds, pending columns, points per call -- all refer to the same launches), `st_all` for every step
        # (counters)
        st, st_plain = M.Stats(), M.Stats()
        iters = 0
        every = max(1, args.timing_every)
        # host wall time of every step (a solve ends with a host wait)
        step_ms, step_timed = [], []
        barrier()
        t0 = time.perf_counter()
        tp = t0
        for i in range(count):
            timed_step = not args.no_kernel_timing and i % every == 0
            r, xx = solve(stats=st if timed_step else st_plain, flags=M.TIME_KERNELS if timed_step else 0, s=s)
            iters += r.iterations
            tn = time.perf_counter()
            step_ms.append((tn - tp) * 1e3)
            step_timed.append(timed_step)
            tp = tn
        barrier()
        dt = time.perf_counter() - t0
        timed.last_steps = (step_ms, step_timed)
        if distributed:
            tt = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        d, dp = st.as_dict(), st_plain.as_dict()
        d["all"] = {k: (d[k] + dp[k]) if not isinstance(d[k], list) else [a + b for a, b in zip(d[k],
                dp[k])] for k in d}
        return d, iters, dt, r, xx

    st, iters, dt, res, x = timed(args.steps, settings)
    main_steps = timed.last_steps
    sta = st["all"]
    timed_steps = len(range(0, args.steps, max(1, args.timing_every))) if not args.no_kernel_timing else 0
    if res.status < 0:
        raise SystemExit(f"solver failed: {res}")
    survey = None
    if args.survey_steps > 0:
        s9 = M.LeastSquaresSettings()
        s9.absTolerance = 1e-9
        solve(s=s9)
        st9, it9, dt9, r9, _ = timed(args.survey_steps, s9)
        p9 = st9["all"]["passes"] / args.survey_steps
        survey = {"abs_tolerance": 1e-9, "value": it9 / dt9, "unit": "iterations/s", "steps": args.survey_steps,
                  "ms_per_solve": dt9 / args.survey_steps * 1e3, "iterations_per_solve": it9 / args.survey_steps,
                  "passes_per_solve": p9, "status": r9.status.name, "residual": r9.residual,
                  # which of the two branches the noise-decided last acceptance took (BASELINE.md
                  # section 2)
                  "branch": ("xConverged after the confirming step (short: ~12-22 passes)"
                             if r9.status.name == "xConverged"
                             else "the confirming step was rejected: the reference's lambda ladder runs to maxLambda "
                                  "(~45 more rejected passes)"),
                  "ms_per_step_min_median_max": step_stats(timed.last_steps[0])}

    out = None
    if rank == 0:
        K = args.steps
        value = iters / dt                                   # iterations of the GLOBAL solve per second, at every N
        nb = max(1, st["jtj_broyden_launches"])
        kern_ms = st["jtj_broyden_ms"] / nb
        survey_bytes = 8.0 * (2.0 * m * n + 3.0 * m)         # T (2 m n + 3 m), Broyden pass with J rewritten
        # broyden_lr.h: J is read once and never written; the sweep also reads the k pending columns of U, y_new, y_old
        # and writes one column: T (m n + (k + 3) m), k averaged over the timed launches
        kbar = st["broyden_lr_columns"] / nb
        alg_bytes = 8.0 * (m * n + (kbar + 3.0) * m)
        ncp = 1 if n <= 32 else 2 if n <= 64 else 4 if n <= 128 else 8
        kname = f"mirlsq::k_broyden_lr<double, {ncp}, true>"
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if st["jtj_broyden_launches"] else 0.0
        sweep = {
            "kernel": kname + " (Broyden pass as a read-only sweep over J: u, J^T u, J^T y, pending rank-one terms)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_field(kname, m, n, "hbm_bytes_per_launch"), "traffic_source": traffic_source(m, n),
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms,
            "launches": st["jtj_broyden_launches"], "pending_columns_avg": kbar,
            "survey_unit_bytes": survey_bytes,             # what the reference's formulation of the pass moves
            "survey_unit_rate_GBs": survey_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms else 0.0,
        }
        nfd = st["jtj_fd_launches"]
        npl = st["jtj_launches"] - st["jtj_broyden_launches"] - nfd
        ncb = (n + 15) // 16
        jtj_flops = m * n * (n + 1.0) + 2.0 * m * n
        if nfd:
            fd_ms = st["jtj_fd_ms"] / nfd
            # the m x n difference panel (fbRowMajorDiff)
            diff_panel = args.fd == "batched" and ((n <= 128 and n % 2 == 0) or n in (192, 256))
            fd_name = (f"mirlsq::k_jtj_fdp<{ncb}, false, true>" if diff_panel else f"mirlsq::k_jtj_fdp<{ncb}, true, "
                                                                                   f"false>") if n <= 128 \
                else f"mirlsq::k_jtj_fdp8<{ncb}, {'true' if diff_panel else 'false'}>"
            # read the panel (m x n differences, or m x 2n pairs) and y, write J
            fd_bytes = 8.0 * ((2.0 if diff_panel else 3.0) * m * n + m)
            fd_rate = fd_bytes / (fd_ms * 1e-3) / 1e9
            fd_tf = (jtj_flops + 2.0 * m * n) / (fd_ms * 1e-3) / 1e12
            # which roofline bounds it: n (n + 1) flop against 16 (or 24) bytes per row element -- at n = 128 the HBM
            # time at 8 TB/s (0.26 ms) exceeds the MFMA time at 78.6 TF (0.21 ms), at n = 256 it is the other way round
            # (0.51 vs 0.84 ms)
            mfma_bound = (jtj_flops / (F64_MFMA_PEAK_TF * 1e12)) > (fd_bytes / (HBM_PEAK_GBS * 1e9))
            fresh = {
                "kernel": fd_name + (" (finite-difference rows from the m x n DIFFERENCE panel" if diff_panel else
                                     " (finite-difference rows from the (+h, -h) pair panel")
                                  + " -> J, J^T J + J^T y on f64 MFMA 16x16x4, register-staged producer waves + MFMA "
                                    "consumer waves)",
                **({"bound": "mfma", "achieved": fd_tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": fd_tf / F64_MFMA_PEAK_TF,
                    "hbm_GBs": fd_rate, "hbm_frac": fd_rate / HBM_PEAK_GBS} if mfma_bound else
                   {"bound": "hbm", "achieved": fd_rate, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": fd_rate / HBM_PEAK_GBS}),
                "traffic": pmc_field(fd_name, m, n, "hbm_bytes_per_launch"), "traffic_source": traffic_source(m, n),
                "algorithmic_bytes_per_launch": fd_bytes, "avg_launch_ms": fd_ms,
                "launches": nfd, "mfma_tflops": (jtj_flops + 2.0 * m * n) / (fd_ms * 1e-3) / 1e12,
                "mfma_util_pmc": pmc_field(fd_name, m, n, "mfma_util"),
            }
        else:
            plain_ms = (st["jtj_ms"] - st["jtj_broyden_ms"]) / max(1, npl)
            pl_name = (f"mirlsq::k_jtj8<{ncb}>" if n > 128 else f"mirlsq::k_jtj_fdp<{ncb}, false>" if n % 2 == 0
                       else f"mirlsq::k_jtj<double, {ncb}, false>")
            tf = jtj_flops / (plain_ms * 1e-3) / 1e12 if plain_ms else 0.0
            fresh = {
                "kernel": pl_name + " (J^T J + J^T y of a fresh Jacobian, f64 MFMA 16x16x4)",
                "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": tf / F64_MFMA_PEAK_TF,
                "avg_launch_ms": plain_ms, "launches": npl,
                "traffic": pmc_field(pl_name, m, n, "hbm_bytes_per_launch"), "traffic_source": traffic_source(m, n),
                "algorithmic_bytes_per_launch": 8.0 * (m * n + m),
                "mfma_util_pmc": pmc_field(pl_name, m, n, "mfma_util"),
            }
        fresh_total = fresh["avg_launch_ms"] * fresh["launches"]
        sweep_total = sweep["avg_launch_ms"] * sweep["launches"]
        dominant = fresh if fresh_total >= sweep_total else sweep                 # the busiest LIBRARY kernel

        # ---- the caller-side kernels (the synthetic workload's residual callbacks), timed by the solver on its stream
        user = {}
        # (two-stream window refreshes overlap the caller's kernels with the library's: not timed apart)
        if st["fd_callback_calls"] and st["fd_callback_ms"] > 0:
            ms = st["fd_callback_ms"] / st["fd_callback_calls"]
            pts = st["fd_callback_points"] / st["fd_callback_calls"]
            if args.fd == "serial":
                by = pts * 8.0 * (m * n + m)
                rate = by / (ms * 1e-3) / 1e9
                user["residual_gemm"] = {"kernel": "wl k_tanh_linear (one sweep over A per finite-difference point)",
                                         "bound": "hbm",
                                         "achieved": rate, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": rate / HBM_PEAK_GBS,
                                         "avg_call_ms": ms, "calls": st["fd_callback_calls"], "points_per_call": pts,
                                         "algorithmic_bytes_per_call": by}
            else:
                fl = 2.0 * m * n * pts                          # A[m x n] . X^T[n x p]; + one tanh per output
                tf = fl / (ms * 1e-3) / 1e12
                diff_panel = args.fd == "batched" and ((n <= 128 and n % 2 == 0) or n in (192, 256))
                by = 8.0 * (m * n + m * pts * (0.5 if diff_panel else 1.0) + m)   # read A once, write the panel
                kn = "k_tanh_linear_batched_dma"
                kn_full = f"k_tanh_linear_batched_dma<{n // 4}, true, {'true' if diff_panel else 'false'}>"
                user["residual_gemm"] = {"kernel": f"wl {kn} (caller side: the 2n finite-difference points as one A . "
                                                   f"X^T GEMM on f64 MFMA "
                                                   "+ tanh epilogue, writes the "
                                                   + ("m x n difference panel)" if diff_panel else "m x 2n panel)"),
                                         "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                         "frac": tf / F64_MFMA_PEAK_TF, "avg_call_ms": ms,
                                             "calls": st["fd_callback_calls"],
                                         "points_per_call": pts, "algorithmic_bytes_per_call": by,
                                         "algorithmic_GBs": by / (ms * 1e-3) / 1e9,
                                         "traffic": pmc_field(kn_full, m, n, "hbm_bytes_per_launch"),
                                         "mfma_util_pmc": pmc_field(kn_full, m, n, "mfma_util"),
                                         "traffic_source": traffic_source(m, n)}
        if st["trial_callback_calls"]:
            ms = st["trial_callback_ms"] / st["trial_callback_calls"]
            pts = st["trial_callback_points"] / st["trial_callback_calls"]
            # one sweep over A serves the points of a call (ladder trials)
            by = 8.0 * (m * n + pts * m + m)
            rate = by / (ms * 1e-3) / 1e9
            user["trial_residual"] = {"kernel": "wl k_tanh_linear / k_tanh_linear_multi (caller side: f(trial), one "
                                                "sweep over A per call)",
                                      "bound": "hbm", "achieved": rate, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": rate / HBM_PEAK_GBS, "avg_call_ms": ms,
                                          "calls": st["trial_callback_calls"],
                                      "points_per_call": pts, "algorithmic_bytes_per_call": by}
        solve_k = {"kernel": "mirlsq::k_lm_solve (damping, posvx('E','L'), BOXCQP, step rounding, prediction: one "
                             "workgroup per ladder entry)",
                   "bound": "latency", "avg_launch_ms": st["solve_ms"] / max(1, st["solve_launches"]),
                   "launches": st["solve_launches"], "flops_per_launch": n ** 3 / 3.0}
        KT = max(1, timed_steps)                             # steps of the timed region whose kernels were event-timed
        lib_ms = (st["jtj_ms"] + st["solve_ms"]) / KT
        user_ms = (st["fd_callback_ms"] + st["trial_callback_ms"]) / KT
        fd_names = {"batched": "batched difference-panel", "rowmajor": "batched pair-panel",
                    "pointmajor": "batched point-major", "serial": "single-point"}
        out = {
            "metric": "LM iterations/sec", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
            # host wall time per step on rank 0 (min, median, max): all K steps, and the steps without kernel events
            # only
            "ms_per_step_min_median_max": step_stats(main_steps[0]),
            "ms_per_step_uninstrumented_min_median_max": step_stats([t for t, e in zip(*main_steps) if not e]),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"cfg3 tanh-linear NLS m_total={m_total} x n={n} fp64, FD Jacobian (central differences, "
                            f"2n residual evaluations per refresh through the "
                            f"{fd_names[args.fd]} callback), "
                            f"absTolerance={args.abs_tolerance:g}, whole solves x0 -> termination, {args.scaling} "
                            f"scaling "
                            f"({m} rows on rank 0)",
                "m_total": m_total, "m_per_gpu": m, "n": n, "scaling": args.scaling,
                "parallelism": (f"ONE rank of {replay} (rows sharded x{replay}; all-reduce totals replayed from the "
                                f"recorded {replay}-shard solve)"
                                if replay else f"rows sharded x{world}, ")
                               + ("" if replay else "RCCL all-reduce" if args.comm == "rccl" else
                                                                 "gloo callback all-reduce ("
                                  + ("FALLBACK: RCCL unusable" if comm_fallback else "rehearsal") + ")"),
                "rccl_ranks": api.lib().mir_lsq_comm_ranks(comm) if (comm and args.comm == "rccl") else None,
                "rccl_fallback_reason": comm_fallback,
                "launcher": ("bench.py (self-launched rank processes)" if os.environ.get("BENCH_SELF_LAUNCHED") == "1"
                             else "external (torch.distributed.run)") if world > 1 else None,
                "ranks_share_gpus": bool(share and world > 1), "visible_gpus": ndev,
                "comm": describe_comm(api, comm),
                    # transport, the shared object RCCL was bound from, its version, ncclCommCount
                # library kernel launches per round, by the kind of round (refresh / Broyden / re-solve after a
                # rejection)
                "library_launches_per_round": {k: (sta["round_launches"][i] / sta["rounds"][i]
                                                   if sta["rounds"][i] else None)
                                               for i, k in enumerate(("refresh", "broyden", "resolve"))},
                "rounds_per_solve": {k: sta["rounds"][i] / K for i, k in enumerate(("refresh", "broyden", "resolve"))},
                "allreduce_per_solve": {"packed_calls": sta["allreduce_calls"][0] / K,
                                        "packed_elems": sta["allreduce_elems"][0] / max(1, sta["allreduce_calls"][0]),
                                        "sweep_calls": sta["allreduce_calls"][1] / K,
                                            "sweep_elems": sta["allreduce_elems"][1] / max(1,
                                                sta["allreduce_calls"][1]),
                                        "scalar_calls": sta["allreduce_calls"][2] / K},
                "rccl_stall_probe": stall if t_comm is not None else None,
                "replay": replay_info,
                "abs_tolerance": args.abs_tolerance,
                "abs_tolerance_note": "1e-5: every accept/reject decision of the solve has margin; at the survey's "
                                      "1e-9 the last "
                                      "acceptance compares rounding noise (12 it / 16 passes or 11 it / 56 passes): "
                                      "see survey_setting",
                "survey_setting": survey,
                "iterations_per_solve": iters / K, "status": res.status.name,
                "passes_per_solve": sta["passes"] / K, "fcalls_per_solve": res.fCalls,
                "jacobian_full_per_solve": sta["jacobian_full"] / K, "residual": res.residual,
                "kernel_timing": f"HIP events on the solver's stream in {timed_steps} of the {K} timed steps (every "
                                 f"{max(1, args.timing_every)}th)",
                "time_split_ms_per_solve": {
                    "caller_fd_callbacks": st["fd_callback_ms"] / KT,
                        "caller_trial_callbacks": st["trial_callback_ms"] / KT,
                    "jtj_fd_kernel": st["jtj_fd_ms"] / KT, "broyden_sweep": st["jtj_broyden_ms"] / KT,
                    "solve_kernel": st["solve_ms"] / KT, "library_kernels": lib_ms, "caller_kernels": user_ms,
                    "total_wall": sta["total_ms"] / K},
            },
            # `roofline` = the kernel with the most time in the TIMED REGION, caller-side kernels included (round-4
            # review: the caller's GEMM is 46 % of the GPU time at cfg 3, the library's busiest kernel 13 %); the two
            # hot library kernels always have their own objects (jtj_kernel, broyden_kernel), the caller's theirs
            # (residual_gemm, trial_residual)
            "roofline": None,
            "jtj_kernel": fresh,
            "broyden_kernel": sweep,
            **user,
            "solve_kernel": solve_k,
        }
        cands = [("library", "jtj_kernel", fresh, fresh_total), ("library", "broyden_kernel", sweep, sweep_total)]
        for key in ("residual_gemm", "trial_residual"):
            if key in user:
                cands.append(("caller", key, user[key], user[key]["avg_call_ms"] * user[key]["calls"]))
        side, key, obj, tot = max(cands, key=lambda c: c[3])
        out["roofline"] = dict(obj, object=key, side=side, total_ms_in_timed_steps=tot,
                               avg_launch_ms=obj.get("avg_launch_ms", obj.get("avg_call_ms")),
                                   launches=obj.get("launches", obj.get("calls")),
                               library_dominant={"object": ("jtj_kernel" if fresh_total >= sweep_total
                                                            else "broyden_kernel"),
                                                 "frac": dominant["frac"], "bound": dominant["bound"]})
        if world == 1 and not args.no_host_callback and (m, n) == (1_000_000, 128):
            # the path a caller of the UNMODIFIED reference API gets: host residual callback, native thread manager,
            # PCIe inclusive
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import bench_host_callback as BH
                hc = BH.run(m, n, abs_tolerance=args.abs_tolerance, data=data, solves=1)
                xh = hc.pop("x")
                hc["parity_x_max_abs_diff_vs_device_callback_solve"] = float(
                    np.abs(np.asarray(xh) - np.asarray(x)).max())
                out["host_callback_mode"] = hc
            except Exception as e:      # noqa: BLE001 -- an auxiliary leg must not take the headline line down with it
                out["host_callback_mode"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, m, n, args.cpu_iterations, args.abs_tolerance,
                                               min(os.cpu_count() or 1, 64), x, res)
            if not args.no_cpu_1thread:
                out["cpu_baseline_1thread"] = cpu_baseline(data, m, n, 1, args.abs_tolerance, 1)
    if comm_obj is not None:
        comm_obj.close()
    elif comm:
        api.lib().mir_lsq_comm_destroy(comm)
    if inner_comm:
        api.lib().mir_lsq_comm_destroy(inner_comm)
    api.lib().mir_lsq_workspace_destroy(ws)
    if distributed:
        dist.destroy_process_group()
    flush_c_stdio()
    if out is not None:
        # the JSON line is the LAST line of the job's stdout: everything is torn down and flushed, and with several
        # ranks the others get a moment to exit first
        if world > 1:
            time.sleep(1.0)
        print(json.dumps(out), flush=True)




def cpu_baseline(data, m, n, iterations, abs_tolerance, threads, x_gpu=None, res_gpu=None):
    """The oracle (port of the reference algorithm; OpenBLAS from scipy for syrk/gemv/ger/posvx -- the library class the
    reference links) on the same inputs, bounded to `iterations` accepted steps, on `threads` host threads."""
    from oracle import oracle as O
    ob = O.load_openblas(threads=threads)
    if ob:
        O.lib().lmo_openblas_set_threads(int(threads))
    omp_set = getattr(O.lib(), "lmo_set_omp_threads", None)
    if omp_set is not None:
        omp_set(int(threads))
    so = O.default_settings()
    so.absTolerance = abs_tolerance
    so.maxIterations = iterations
    ctx = O.TanhLinearCtx(data["A"].ctypes.data, data["b"].ctypes.data)
    t0 = time.perf_counter()
    ro, xo = O.optimize(O.native_fn("wlc_tanh_linear_f"), m, data["x0"], settings=so, fctx=C.addressof(ctx),
                        use_openblas=ob)
    dt = time.perf_counter() - t0
    parity = {}
    if x_gpu is not None and ro.iterations == res_gpu.iterations:
        # the CPU sample ran the same number of accepted iterations as the GPU solve: the full-size parity datum
        import numpy as np
        parity = {"parity_x_max_abs_diff": float(np.abs(np.asarray(x_gpu) - np.asarray(xo)).max()),
                  "parity_x_max_abs": float(np.abs(np.asarray(xo)).max()),
                  "parity_residual_rel_diff": abs(res_gpu.residual - ro.residual) / abs(ro.residual),
                  "parity_status": [int(res_gpu.status), int(ro.status)]}
    return {**parity, "value": ro.iterations / dt, "unit": "iterations/s", "cores": threads,
            "host_nproc": os.cpu_count(),
            "kind": "port",
            "sample": f"the first {ro.iterations} accepted LM iteration(s) of the same m={m} x n={n} solve (bounded by "
                      f"maxIterations={iterations}; status {O.STATUS.get(ro.status, ro.status)}, "
                      f"fCalls {ro.fCalls}: FD Jacobians of {2 * n} residual calls each + Broyden passes), {dt:.1f} s, "
                      f"OpenBLAS={'yes' if ob else 'no (plain loops)'} x{threads}, residual calls OpenMP x{threads}",
            "seconds": dt, "fcalls": ro.fCalls}
