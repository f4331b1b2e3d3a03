#!/usr/bin/env python3
"""bench.py -- LM iterations/sec on the BASELINE.json headline workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload (cfg 3): tanh-linear synthetic NLS, r_i(x) = tanh(a_i . x) - b_i,
m = 1e6 rows PER GPU x n = 128 parameters, fp64, finite-difference Jacobian through the user
residual callback, defaults except absTolerance (1e-5: every decision of the solve has margin; at the survey's 1e-9
the last acceptance compares rounding noise and the pass count is a coin flip); rank r owns global rows
[r * 1e6, (r + 1) * 1e6) (weak scaling; one fused RCCL all-reduce of [J^T J | J^T y] per
Jacobian-changing pass, one scalar all-reduce per trial step).

A "step" is one complete LM solve (mir_optimize_least_squares_gpu_d from x0 to termination):
residual + FD Jacobian callbacks, Broyden updates, J^T J / J^T y, damped BOXCQP solves, step
acceptance -- nothing is skipped or cached between solves. `value` = (accepted LM iterations of the
K timed solves) x N / time: LM iterations per second per 1e6 x 128 row block, aggregated over ranks
(at N = 1 it is plainly the solver's LM iterations/sec).

The JSON line also carries
  roofline     -- the library kernel with the most time in the timed region, HIP-event timed on the solver's stream:
                  k_jtj_fdp (finite-difference rows -> J, J^T J + J^T y on f64 MFMA; HBM-bound, algorithmic
                  bytes 8 (3 m n + m)) or the Broyden sweep k_broyden_lr (HBM-bound, 8 (m n + (k + 3) m))
  broyden_kernel / jtj_kernel -- the other one of the two
  cpu_baseline -- the oracle (CPU port of the reference algorithm, OpenBLAS for syrk/gemv/ger/posvx)
                  on a bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_field(kernel, m, n, field):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/rNN/pmc_traffic.json,
    made by scripts/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command).
    PMC counters cannot be read from inside the timed run, so this is the stored measurement; None if absent or
    if the run's shape differs from the profiled one (m = 1e6, n = 128)."""
    import glob
    if (m, n) != (1_000_000, 128):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))["kernels"][kernel][field]
    except (KeyError, ValueError):
        return None


def pmc_traffic(kernel, m, n):
    return pmc_field(kernel, m, n, "hbm_bytes_per_launch")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    # (under torch.distributed.run, abbreviations such as --m collide with the launcher's own options: BENCH_M / BENCH_N)
    ap.add_argument("--m", type=int, default=int(os.environ.get("BENCH_M", 1_000_000)), help="rows per GPU")
    ap.add_argument("--n", type=int, default=int(os.environ.get("BENCH_N", 128)))
    ap.add_argument("--fd", choices=["batched", "pointmajor", "serial"], default="batched",
                    help="finite differences through the batched residual callbacks (row-major panel, fill fused into the "
                         "J^T J kernel), through the point-major batched callback + k_fd_fill, or one call per point")
    ap.add_argument("--abs-tolerance", type=float, default=1e-5,
                    help="LeastSquaresSettings.absTolerance of the workload (see DESIGN.md section 5 for why not 1e-9)")
    ap.add_argument("--control-plane", choices=["gloo", "nccl"], default="gloo",
                    help="torch.distributed backend for barriers / id exchange (the solve's collectives always use the "
                         "solver's own RCCL communicator)")
    ap.add_argument("--comm", choices=["rccl", "gloo-callback"], default="rccl",
                    help="data-plane communicator: the solver's RCCL communicator (production) or, to rehearse N > 1 on a box "
                         "with fewer GPUs, its callback communicator over gloo (ranks then share GPUs)")
    ap.add_argument("--settle", type=float, default=8.0,
                    help="N > 1: minimum seconds between RCCL communicator creation and the first warm-up solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iterations", type=int, default=6, help="accepted iterations of the CPU sample")
    return ap.parse_args()


def flush_c_stdio():
    """RCCL prints its banner through C stdio, which is fully buffered on a pipe."""
    C.CDLL(None).fflush(None)
    sys.stdout.flush()


def main():
    args = parse()
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))   # CPU baseline leg (oracle, OpenMP)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    import numpy as np
    import torch
    import torch.distributed as dist

    import mir_optim_amd as M
    from mir_optim_amd import api, workloads as W

    if not torch.cuda.is_available() or M.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: mir_optim_amd has no CPU path")
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if args.comm == "gloo-callback" else local_rank)
    comm = None
    comm_obj = None
    distributed = world > 1 or os.environ.get("MIR_LSQ_FORCE_COMM") == "1"   # the env knob exercises the N > 1 code path on one GPU
    if distributed:
        from mir_optim_amd import parallel as PAR
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # Control plane (unique-id exchange, the barriers around the timed region, the max over ranks): torch.distributed.
        # Data plane (every collective of the solve): the solver's OWN RCCL communicator over xGMI, created below.
        # The control plane defaults to gloo: a second, idle RCCL instance (torch's "nccl" process group with its
        # watchdog / heartbeat threads and streams) next to the solver's communicator buys nothing;
        # --control-plane nccl selects it anyway.
        if args.control_plane == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        ctl_dev = "cuda" if args.control_plane == "nccl" else "cpu"

        def bcast(buf):
            t = torch.from_numpy(buf).to(ctl_dev)
            dist.broadcast(t, 0)
            return t.cpu().numpy()
        if args.comm == "gloo-callback":
            # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks: the solver's callback communicator, its
            # all-reduce done by torch.distributed over gloo (device -> host -> gloo -> device); ranks may share a GPU
            comm_obj = PAR.HostAllreduceComm(world, rank, PAR.torch_allreduce_numpy(dist))
            comm = comm_obj.handle
        elif os.environ.get("BENCH_DIAG_NO_RCCL") != "1":   # diagnostic: process group only
            comm = PAR.rccl_comm(world, rank, bcast)  # the solver's own RCCL communicator (xGMI), id via torch.distributed
        t_comm = time.perf_counter()

    m, n = args.m, args.n
    data = W.tanh_linear_data(m, n, row_offset=rank * m)
    prob = W.TanhLinear(data["A"], data["b"])
    settings = M.LeastSquaresSettings()
    settings.absTolerance = args.abs_tolerance
    ws = api.lib().mir_lsq_workspace_create(m, n, 8)
    if not ws:
        raise SystemExit("workspace allocation failed")

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def solve(stats=None, flags=0):
        return prob.solve(data["x0"], settings=settings, stats=stats, flags=flags, comm=comm, workspace=ws,
                          batched={"batched": True, "pointmajor": "pointmajor", "serial": False}[args.fd])

    # RCCL finishes part of its initialisation asynchronously: a few seconds after ncclCommInitRank (of the solver's
    # communicator; torch's process group alone does not show it) every HIP launch of the process stalls once or twice for
    # 60-150 ms. Measured on the one-GPU box with MIR_LSQ_FORCE_COMM=1: 160-310 it/s when that lands in the timed region,
    # 740-760 when it does not; warm collectives at creation and RCCL_MSCCL*/NCCL_* knobs do not move it, waiting does.
    # So: keep `settle` seconds between communicator creation and the first warm-up solve (set-up time counts).
    if comm is not None and args.comm == "rccl" and args.settle > 0:
        wait = args.settle - (time.perf_counter() - t_comm)
        if wait > 0:
            time.sleep(wait)
    for _ in range(args.warmup):
        res, x = solve()
    flush_c_stdio()     # every rank: RCCL's init banner leaves the C stdio buffer now, not at process exit
    stats = M.Stats()
    iters = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, x = solve(stats=stats, flags=M.TIME_KERNELS)
        iters += res.iterations
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if res.status < 0:
        raise SystemExit(f"solver failed: {res}")

    out = None
    if rank == 0:
        st = stats.as_dict()
        value = iters * world / dt
        nb = max(1, st["jtj_broyden_launches"])
        kern_ms = st["jtj_broyden_ms"] / nb
        survey_bytes = 8.0 * (2.0 * m * n + 3.0 * m)         # T (2 m n + 3 m), Broyden pass with J rewritten
        lowrank = os.environ.get("MIR_LSQ_BROYDEN", "")[:1] != "f"
        if lowrank:
            # broyden_lr.h: J is read once and never written; the sweep also reads the k pending columns of U,
            # y_new, y_old and writes one column: T (m n + (k + 3) m), k averaged over the timed launches
            kbar = st["broyden_lr_columns"] / nb
            alg_bytes = 8.0 * (m * n + (kbar + 3.0) * m)
            ncp = 1 if n <= 32 else 2 if n <= 64 else 4 if n <= 128 else 8
            kname = f"mirlsq::k_broyden_lr<double, {ncp}, true>"
            kdesc = kname + " (Broyden pass as a read-only sweep over J: u, J^T u, J^T y, pending rank-one terms)"
            kflops = 6.0 * m * n
        else:
            kbar = 0.0
            alg_bytes = survey_bytes
            kname = "mirlsq::k_jtj2<8, true>"
            kdesc = kname + " (fused Broyden + J^T J + J^T y, LDS-DMA ring, J rewritten)"
            kflops = m * n * (n + 1.0) + 6.0 * m * n
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if st["jtj_broyden_launches"] else 0.0
        sweep = {
            "kernel": kdesc, "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(kname, m, n), "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms,
            "launches": st["jtj_broyden_launches"], "pending_columns_avg": kbar,
            "survey_unit_bytes": survey_bytes,             # what the reference's formulation of the pass moves
            "survey_unit_rate_GBs": survey_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms else 0.0,
            "valu_tflops": kflops / (kern_ms * 1e-3) / 1e12 if kern_ms else 0.0,
        }
        # J^T J + J^T y of a refreshed Jacobian on f64 MFMA: k_jtj2<., false> after k_fd_fill, or, with the row-major
        # batched callback, k_jtj2<., false, true> which also forms the Jacobian rows from the FD panel and writes J
        nfd = st["jtj_fd_launches"]
        npl = st["jtj_launches"] - st["jtj_broyden_launches"] - nfd
        ncb = (n + 15) // 16
        jtj_flops = m * n * (n + 1.0) + 2.0 * m * n
        if nfd:
            fd_ms = st["jtj_fd_ms"] / nfd
            ring = os.environ.get("MIR_LSQ_FD_KERNEL", "")[:1] == "r"
            fd_name = f"mirlsq::k_jtj2<{ncb}, false, true>" if ring else f"mirlsq::k_jtj_fdp<{ncb}, true>"
            fd_bytes = 8.0 * (3.0 * m * n + m)                # read the m x 2n panel and y, write J
            fd_rate = fd_bytes / (fd_ms * 1e-3) / 1e9
            fresh = {
                "kernel": fd_name + " (finite-difference rows from the (+h, -h) panel -> J, J^T J + J^T y on f64 MFMA 16x16x4, "
                                    + ("LDS-DMA ring)" if ring else "register-staged producer waves + MFMA consumer waves)"),
                "bound": "hbm", "achieved": fd_rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fd_rate / HBM_PEAK_GBS,
                "traffic": pmc_traffic(fd_name, m, n), "algorithmic_bytes_per_launch": fd_bytes, "avg_launch_ms": fd_ms,
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
                "bound": "mfma", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                "avg_launch_ms": plain_ms, "launches": npl,
                "traffic": pmc_traffic(pl_name, m, n), "algorithmic_bytes_per_launch": 8.0 * (m * n + m),
                # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) from the committed PMC pass: pipe occupancy
                # at the clock the kernel actually ran at (the TFLOP/s fraction above is against the 2.4 GHz peak)
                "mfma_util_pmc": pmc_field(pl_name, m, n, "mfma_util"),
            }
        # `roofline` is the library kernel with the most time in the timed region; the other one rides along
        fresh_total = fresh["avg_launch_ms"] * fresh["launches"]
        sweep_total = sweep["avg_launch_ms"] * sweep["launches"]
        dominant, other, other_key = (fresh, sweep, "broyden_kernel") if fresh_total >= sweep_total else (sweep, fresh, "jtj_kernel")
        out = {
            "metric": "LM iterations/sec", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"cfg3 tanh-linear NLS m={m}/GPU x n={n} fp64, FD Jacobian ({args.fd} residual callback), "
                            f"absTolerance={args.abs_tolerance:g}, whole solves x0 -> termination",
                "m_per_gpu": m, "m_total": m * world, "n": n, "parallelism": f"rows sharded x{world}, " + ("RCCL all-reduce" if args.comm == "rccl" else "gloo callback all-reduce (rehearsal)"),
                "abs_tolerance": args.abs_tolerance,
                "abs_tolerance_note": "1e-5: every accept/reject decision of the solve has margin; at 1e-9 the last acceptance "
                                      "compares rounding noise (12 it / 16 passes or 11 it / 56 passes), DESIGN.md section 5",
                "iterations_per_solve": iters / args.steps, "status": res.status.name,
                "passes_per_solve": st["passes"] / args.steps, "fcalls_per_solve": res.fCalls,
                "jacobian_full_per_solve": st["jacobian_full"] / args.steps,
                "global_lm_iterations_per_sec": iters / dt, "residual": res.residual,
                "time_split_ms_per_solve": {
                    "fd_refresh": st["fd_ms"] / args.steps, "jtj_kernels": st["jtj_ms"] / args.steps,
                    "solve_kernel": st["solve_ms"] / args.steps, "total": st["total_ms"] / args.steps},
            },
            "roofline": dominant,
            other_key: other,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, m, n, args.cpu_iterations, args.abs_tolerance, x, res)
    if comm_obj is not None:
        comm_obj.close()
    elif comm:
        api.lib().mir_lsq_comm_destroy(comm)
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


def cpu_baseline(data, m, n, iterations, abs_tolerance, x_gpu=None, res_gpu=None):
    """The oracle (port of the reference algorithm; OpenBLAS from scipy for syrk/gemv/ger/posvx -- the
    library class the reference links) on the same inputs, bounded to `iterations` accepted steps."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    ob = O.load_openblas(threads=threads)
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
    return {**parity, "value": ro.iterations / dt, "unit": "iterations/s", "cores": threads, "kind": "port",
            "sample": f"the first {ro.iterations} accepted LM iterations of the same m={m} x n={n} solve (bounded by "
                      f"maxIterations={iterations}; status {O.status_name(ro.status) if hasattr(O, 'status_name') else ro.status}, "
                      f"fCalls {ro.fCalls}: FD Jacobians of {2 * n} residual calls each + Broyden passes), {dt:.1f} s, "
                      f"OpenBLAS={'yes' if ob else 'no (plain loops)'}, residual calls OpenMP x{threads}",
            "seconds": dt, "fcalls": ro.fCalls}


if __name__ == "__main__":
    main()
