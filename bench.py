#!/usr/bin/env python3
"""bench.py -- LM iterations/sec on the BASELINE.json headline workload.

  python bench.py --gpus N --steps K --warmup W
  N > 1 from a bare shell: this process starts the N rank processes itself (launch_ranks: children of this interpreter with
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; the parent never imports torch or touches HIP) and relays rank 0's JSON line.
  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (WORLD_SIZE set) it is one rank.

Workload (cfg 3): tanh-linear synthetic NLS, r_i(x) = tanh(a_i . x) - b_i, m = 1e6 rows x n = 128
parameters, fp64, finite-difference Jacobian through the user's batched residual callbacks, defaults except absTolerance
(1e-5: every accept / reject decision of the solve has margin; the survey's 1e-9 is measured too and reported in
config.survey_setting -- there the last acceptance compares rounding noise).

Scaling (BASELINE.json: "m=1e6 x n=128 ... 1/2/4/8 MI355X"):
  --scaling strong (default)  the SAME 1e6-row problem, rows split over the N ranks (rank r owns rows
                              row_shard(1e6, N, r)); `value` = LM iterations of that one global solve per second.
  --scaling weak              1e6 rows PER GPU (cfg 4's partition; the global problem grows with N); `value` is still the
                              global solve's iterations per second -- it does NOT multiply by N.
Per pass the ranks exchange one sum all-reduce of the packed [J^T J | J^T y] (full refresh) or of the 2n + 34 sweep vector
(Broyden pass) and one of the trial residual sums, on the solver's own RCCL communicator.

A "step" is one complete LM solve (mir_optimize_least_squares_gpu_d from x0 to termination): residual + FD Jacobian
callbacks, Broyden updates, J^T J / J^T y, damped BOXCQP solves, step acceptance -- nothing is skipped or cached between
solves. Inputs are resident in HBM before the timed region.

The JSON line also carries
  roofline       the kernel with the most time in the timed region -- caller-side kernels included (at cfg 3 it is the caller's
                 finite-difference GEMM) --, HIP-event timed on the solver's stream; `object` names the entry it copies
  jtj_kernel, broyden_kernel    the two hot LIBRARY kernels (fused FD / plain J^T J; the Broyden sweep)
  residual_gemm, trial_residual the CALLER-side device callbacks (the synthetic workload's kernels, csrc/workloads.hip + workloads_gemm.hip),
                                event-timed on the same stream: they are most of a solve and get their own roofline objects
  solve_kernel   the one-workgroup n x n kernel (latency-bound; time only)
  cpu_baseline   the oracle (CPU port of the reference algorithm, OpenBLAS for syrk/gemv/ger/posvx) on a bounded sample of
                 the same workload at min(nproc, 64) threads, plus cpu_baseline_1thread; rank 0, N = 1 only.
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
F64_MFMA_PEAK_TF = 78.6        # dense f64 MFMA (= f64 vector) peak


def _round_no(path):
    import re
    m_ = re.search(r"profiles/r(\d+)/", path.replace(os.sep, "/"))
    return int(m_.group(1)) if m_ else -1


def pmc_file(m=1_000_000, n=128):
    """The newest (by round NUMBER) committed PMC summary for the per-GPU shape: profiles/rNN/pmc_traffic.json was taken at
    m = 1e6 x n = 128 (cfg 3), profiles/rNN/n256_pmc.json at m = 1e6 x n = 256 (cfg 4's per-GPU shape)."""
    import glob
    name = {(1_000_000, 128): "pmc_traffic.json", (1_000_000, 256): "n256_pmc.json"}.get((m, n))
    if name is None:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), key=_round_no)
    return files[-1] if files else None


def csrc_sha16(name):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "mir_optim_amd", "csrc", name), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_field(kernel, m, n, field):
    """HBM bytes per launch (or MFMA pipe utilisation) of `kernel` from the COMMITTED rocprofv3 PMC summary (made by
    scripts/pmc_summary.py / pmc_summary2.py from separate --pmc passes of this same command): PMC counters cannot be read
    from inside the timed run, so this is a stored measurement -- `traffic_source` in the JSON line says so. None if absent
    or if the per-GPU shape is not one of the profiled ones (m = 1e6 with n = 128 or 256)."""
    f = pmc_file(m, n)
    if f is None:
        return None
    try:
        ks = json.load(open(f))["kernels"]
        if kernel not in ks:            # template arguments added or dropped since (k_broyden_lr<double, 4, true> <-> <..., true, false>)
            stem = kernel[:-1]
            kernel = next(k for k in ks if k.startswith(stem + ",") or stem.startswith(k[:-1] + ","))
        return ks[kernel][field]
    except (KeyError, ValueError, StopIteration):
        return None


def traffic_source(m, n):
    f = pmc_file(m, n)
    if f is None:
        return None
    return os.path.relpath(f, ROOT) + " (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["cfg3", "cfg2", "cfg5"], default="cfg3",
                    help="cfg3: the headline (BASELINE.json metric). cfg2: BASELINE config 2, Gaussian-sum fit m = 1e5 x n = 16 fp64 "
                         "(launch-latency bound). cfg5: BASELINE config 5, 4096 independent fp32 fits of m = 512 x n = 8, one "
                         "wavefront per problem. cfg2 / cfg5 print additional lines (one GPU; replicas only at N > 1)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    # (under torch.distributed.run, abbreviations such as --m collide with the launcher's own options: BENCH_M / BENCH_N)
    ap.add_argument("--rows", type=int, default=int(os.environ.get("BENCH_M", 1_000_000)),
                    help="rows of the global problem (strong) or per GPU (weak)")
    ap.add_argument("--n", type=int, default=int(os.environ.get("BENCH_N", 128)))
    ap.add_argument("--scaling", choices=["strong", "weak"], default=os.environ.get("BENCH_SCALING", "strong"))
    ap.add_argument("--fd", choices=["batched", "rowmajor", "pointmajor", "serial"], default="batched",
                    help="finite differences through the batched residual callbacks -- batched: the m x n row-major DIFFERENCE "
                         "panel (n <= 128; the caller's kernel subtracts the (+h, -h) pair, the library's fused kernel scales, writes J "
                         "and forms J^T J); rowmajor: the m x 2n row-major pair panel, same fused kernel; pointmajor: the point-major "
                         "batched callback + k_fd_fill -- or one call per point (serial)")
    ap.add_argument("--gemm-read-a-once", action="store_true",
                    help="caller side, --fd batched: the difference-panel GEMM sweeps A once (stage-outer variant: 2.1 instead of "
                         "3.1 GB per call at cfg 3, ~3 %% slower -- the kernel is MFMA-bound; A/B only)")
    ap.add_argument("--abs-tolerance", type=float, default=1e-5,
                    help="LeastSquaresSettings.absTolerance of the headline number (DESIGN.md section 5)")
    ap.add_argument("--survey-steps", type=int, default=10,
                    help="solves timed at SURVEY 8d's absTolerance = 1e-9 after the main region (0 = skip)")
    ap.add_argument("--control-plane", choices=["gloo", "nccl"], default="gloo",
                    help="torch.distributed backend for barriers / id exchange (the solve's collectives always use the "
                         "solver's own RCCL communicator)")
    ap.add_argument("--comm", choices=["rccl", "gloo-callback"], default="rccl",
                    help="data-plane communicator: the solver's RCCL communicator (production) or, to rehearse N > 1 on a box "
                         "with fewer GPUs, its callback communicator over gloo (ranks then share GPUs)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N = 1: attach a one-rank RCCL communicator anyway (exercises the N > 1 code path on one GPU)")
    ap.add_argument("--replay-ranks", type=int, default=0,
                    help="N = 1 only: measure ONE rank of an R-rank strong-scaled job on this GPU. The R-shard solve of the global "
                         "problem runs once (in-process group) and records rank 0's all-reduce totals; every timed solve is then "
                         "rank 0's shard alone, each exchange replaced by the recorded total (mir_lsq_comm_create_replay) -- the "
                         "global trajectory, the rank's own kernels uncontended, no xGMI hop. With --force-comm every exchange "
                         "also passes through a one-rank ncclAllReduce")
    ap.add_argument("--stall-bound", type=float, default=8.0,
                    help="RCCL only: upper bound (s, from communicator creation) of the warm-up loop that waits for RCCL's "
                         "asynchronous initialisation stall to pass (it ends as soon as a stall has been seen and has passed)")
    ap.add_argument("--variant", type=int, default=0, help="MIR_LSQ_VARIANT_* bits for A/B runs (0 = product path)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region (A/B of the instrumentation overhead; the "
                         "roofline objects are then empty)")
    ap.add_argument("--timing-every", type=int, default=10,
                    help="bracket the kernels with HIP events in every k-th step of the timed region (an event record costs a "
                         "few microseconds on the stream: ~0.2 ms per cfg-3 solve when every step is instrumented)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cfg5-replicas", type=int, default=16,
                    help="--config cfg5: also time ONE launch of this many copies of the 4096 problems (the steady-state rate of the kernel, "
                         "where the tail of long fits is amortised); 1 = skip")
    ap.add_argument("--no-host-callback", action="store_true",
                    help="skip the reference-ABI leg (host residual callback + native thread manager, PCIe inclusive; rank 0, N = 1): "
                         "one untimed + one timed solve, a few seconds of host work")
    ap.add_argument("--no-cpu-1thread", action="store_true")
    ap.add_argument("--cpu-iterations", type=int, default=6, help="accepted iterations of the CPU sample")
    return ap.parse_args()


def flush_c_stdio():
    """RCCL prints its banner through C stdio, which is fully buffered on a pipe."""
    C.CDLL(None).fflush(None)
    sys.stdout.flush()


def main_cfg5(args):
    """BASELINE cfg 5: 4096 x (m = 512, n = 8) fp32, one wavefront per problem, the whole LM loop inside ONE kernel launch
    (csrc/batched_kernel.h). A step = one launch = 4096 complete fits from their starting points; inputs resident in HBM.
    value = accepted LM iterations (summed over the problems) per second. Independent problems: N > 1 would be replicas."""
    import numpy as np
    import torch

    import mir_optim_amd as M
    from mir_optim_amd import api
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import problems as P

    if not torch.cuda.is_available() or M.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: mir_optim_amd has no CPU path")
    count, m, n = 4096, 512, 8
    t, data, truth, x0 = P.cfg5_pad8(count, m)
    L = api.lib()
    s = M.LeastSquaresSettings(np.float32)
    dt_, dd, dx0 = api.DeviceBuffer(t), api.DeviceBuffer(data), api.DeviceBuffer(x0)
    dx = api.DeviceBuffer(x0)
    dlo = api.DeviceBuffer(np.full(n, -np.inf, dtype=np.float32))
    dup = api.DeviceBuffer(np.full(n, np.inf, dtype=np.float32))
    dres = api.DeviceBuffer(nbytes=count * 24, dtype=np.uint8, shape=(count * 24,))
    stream = api.Stream()
    # the per-row basis table of the model (4096 x ... no: t is shared, 512 rows x 4 floats) is the caller's: no allocation per launch
    basis = api.DeviceBuffer(nbytes=m * 4 * 4, dtype=np.uint8, shape=(m * 16,))
    bopt = api.BatchedOptions(stream=stream.handle, basis=basis.ptr, basis_bytes=m * 16)

    def step():
        # x is restored on the device (a D2D copy of 128 KB inside the timed region: part of "from the starting points")
        if L.mir_lsq_memcpy_d2d(dx.ptr, dx0.ptr, count * n * 4, stream.handle) != 0:
            raise SystemExit("d2d failed")
        rc = L.mir_lsq_batched_kernel_s(C.byref(s), count, m, M.MODEL_EXP_DECAY_PAD8, dx.ptr, dlo.ptr, dup.ptr, dt_.ptr, 0,
                                        dd.ptr, dres.ptr, C.byref(bopt))
        if rc != 0:
            raise SystemExit(f"batched kernel launch failed: {rc}")
    for _ in range(max(1, args.warmup)):
        step()
    stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    stream.synchronize()
    dt = time.perf_counter() - t0
    raw = np.frombuffer(dres.download().tobytes(), dtype=np.dtype([("status", "<i4"), ("iterations", "<u4"), ("fCalls", "<u4"),
                                                                   ("gCalls", "<u4"), ("residual", "<f4"), ("lambda", "<f4")]))
    iters = int(raw["iterations"].sum())
    fcalls = int(raw["fCalls"].sum())
    ms = dt / args.steps * 1e3
    # Work of one launch: every residual evaluation is m model evaluations (1 exp, ~20 flops; the four sin/cos values of a row do
    # not depend on the parameters and come from the basis table k_batched_basis fills once per launch); a finite-difference
    # Jacobian makes 2 n of them but fCalls counts n (quirk Q5), so 2 x fCalls x m bounds the evaluations from above
    evals = 2.0 * fcalls * m
    out = {
        "metric": "LM iterations/sec", "value": iters / (ms * 1e-3), "unit": "iterations/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cfg5: {count} independent fits m={m} x n={n} fp32, exp-decay family padded to n=8, one wavefront per "
                               "problem, whole LM loop in one kernel launch, FD Jacobian (jacobianEpsilon=2^-11)",
                   "fits_per_s": count / (ms * 1e-3), "iterations_per_fit": iters / count, "fcalls_per_fit": fcalls / count,
                   "status_counts": {str(int(k)): int(v) for k, v in zip(*np.unique(raw["status"], return_counts=True))},
                   "mean_residual": float(raw["residual"].mean()), "parallelism": "replicas only (independent problems)"},
        "roofline": cfg5_roofline(ms, evals, args.steps, count, m, n),
    }
    # ---- steady state (round-4 review): the 4096 fits differ 3 x in length and go to 2048 wave slots, so the launch ends with
    # its stragglers. The same problems 16 times over (65 536 fits in one launch) amortise that tail: the kernel's rate where the
    # dispatcher always has a next problem for a finished wave.
    reps = max(1, args.cfg5_replicas)
    if reps > 1:
        big = count * reps
        dd2, dx02 = api.DeviceBuffer(np.tile(data, (reps, 1))), api.DeviceBuffer(np.tile(x0, (reps, 1)))
        dx2 = api.DeviceBuffer(np.tile(x0, (reps, 1)))
        dres2 = api.DeviceBuffer(nbytes=big * 24, dtype=np.uint8, shape=(big * 24,))

        def step2():
            if L.mir_lsq_memcpy_d2d(dx2.ptr, dx02.ptr, big * n * 4, stream.handle) != 0:
                raise SystemExit("d2d failed")
            if L.mir_lsq_batched_kernel_s(C.byref(s), big, m, M.MODEL_EXP_DECAY_PAD8, dx2.ptr, dlo.ptr, dup.ptr, dt_.ptr, 0, dd2.ptr,
                                          dres2.ptr, C.byref(bopt)) != 0:
                raise SystemExit("batched kernel launch failed")
        step2()
        stream.synchronize()
        k2 = max(3, args.steps // 8)
        t0 = time.perf_counter()
        for _ in range(k2):
            step2()
        stream.synchronize()
        ms2 = (time.perf_counter() - t0) / k2 * 1e3
        raw2 = np.frombuffer(dres2.download().tobytes(), dtype=raw.dtype)
        same = bool((raw2["iterations"].reshape(reps, count) == raw["iterations"][None, :]).all()
                    and (raw2["residual"].view(np.uint32).reshape(reps, count) == raw["residual"].view(np.uint32)[None, :]).all())
        rf = out["roofline"]
        ss = {"fits_per_launch": big, "ms_per_launch": ms2, "fits_per_s": big / (ms2 * 1e-3), "iterations_per_s": iters * reps / (ms2 * 1e-3),
              "speedup_over_4096_fit_launches": (big / ms2) / (count / ms), "replicas_bit_identical_with_the_4096_fit_launch": same}
        if rf.get("valu_instructions_per_launch"):
            ss["valu_frac"] = rf["valu_instructions_per_launch"] * reps / (ms2 * 1e-3) / 1e9 / rf["peak"]
            ss["note"] = ("valu_frac = VALU instructions (the committed count of a 4096-fit launch x replicas: the same problems execute the same "
                          "instructions) / launch time / the issue peak; where the straggler tail is amortised")
        out["config"]["steady_state"] = ss
    if not args.no_cpu_baseline:
        from oracle import oracle as O

        class Ctx(C.Structure):
            _fields_ = [("t", C.c_void_p), ("data", C.c_void_p)]
        f = O.native_fn("wlc_exp_pad8_f_s")
        sample = 1024
        t1 = time.perf_counter()
        it_cpu = 0
        for k in range(sample):
            d = np.ascontiguousarray(data[k])
            ctx = Ctx(t.ctypes.data, d.ctypes.data)
            ro, _ = O.optimize(f, m, x0[k], dtype=np.float32, fctx=C.addressof(ctx))
            it_cpu += ro.iterations
        dtc = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": it_cpu / dtc, "unit": "iterations/s", "cores": 1, "host_nproc": os.cpu_count(), "kind": "port",
                               "sample": f"the first {sample} of the {count} problems, float oracle (plain loops; BLAS has nothing to do at "
                                         f"n = 8), one thread, {dtc:.1f} s incl. ctypes call overhead", "fits_per_s": sample / dtc}
    print(json.dumps(out), flush=True)


def main_cfg2(args):
    """BASELINE cfg 2: Gaussian-sum curve fit, m = 1e5 residuals x n = 16 parameters, fp64, width bounds, FD Jacobian through
    the device callbacks (--fd batched: one launch for the 2n points of a refresh; --fd serial: one per point). J is 12.8 MB: every kernel of a pass is a few microseconds, so the solve is
    bound by launch latency and host round trips, not by HBM or MFMA -- the line reports the time per pass and per launch."""
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
    fdb = {"batched": True, "rowmajor": "rowmajor", "pointmajor": "pointmajor", "serial": False}[args.fd]   # 2n FD points per launch, or one call per point
    for _ in range(max(1, args.warmup)):
        res, x = prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, variant=args.variant, batched=fdb)
    # the timed region carries NO kernel events: at ~6 event pairs per round and 41 rounds per solve they cost 0.9 ms of a
    # 3 ms solve (scripts/ab_bench.sh); the per-kernel split comes from a second, instrumented pass of the same solves
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
        prob.solve(g["x0"], g["lower"], g["upper"], workspace=ws, stats=st, variant=args.variant, flags=M.TIME_KERNELS, batched=fdb)
    d = st.as_dict()
    K = args.steps
    m, n = g["m"], g["n"]
    rounds = d["solve_launches"] / K
    out = {
        "metric": "LM iterations/sec", "value": iters / dt, "unit": "iterations/s", "n_gpus": 1, "steps": K, "warmup": args.warmup,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"cfg2 Gaussian-sum fit m={m} x n={n} fp64, width bounds, FD Jacobian ("
                               + ("batched device callback: the 2n points of a refresh in one launch" if fdb else "single-point device callback")
                               + "), default settings, whole solves",
                   "iterations_per_solve": iters / K, "passes_per_solve": d["passes"] / K, "rounds_per_solve": rounds,
                   "fcalls_per_solve": res.fCalls, "status": res.status.name, "residual": res.residual,
                   "qp_active_set_passes_per_solve": d["qp_active_set_passes"] / K,
                   "us_per_round": dt / K / max(1.0, rounds) * 1e6,
                   "time_split_ms_per_solve": {"caller_fd_callbacks": d["fd_callback_ms"] / K, "caller_trial_callbacks": d["trial_callback_ms"] / K,
                                               "jtj_kernels": d["jtj_ms"] / K, "solve_kernel": d["solve_ms"] / K,
                                               "total_wall_instrumented_pass": d["total_ms"] / K,
                                               "note": "from a second, event-instrumented pass (the timed region has no events)"},
                   "parallelism": "replicas only at N > 1 (the problem is too small to shard)"},
        "roofline": {"kernel": "mirlsq::k_lm_solve<double, 1, true> (the n = 16 damped BOXCQP solve; the longest library kernel of a round)",
                     "bound": "latency", "achieved": None, "peak": None, "unit": "us", "frac": None,
                     "avg_launch_ms": d["solve_ms"] / max(1, d["solve_launches"]), "launches": d["solve_launches"], "traffic": None,
                     "note": "launch-latency bound: J^T J at n = 16 is 2 flop/B (SURVEY 8d) and J is 12.8 MB -- every kernel of a "
                             "round runs for microseconds; the figure of merit is us_per_round"},
    }
    # ---- the resident-J path (include/mir_optim_amd_resident.hpp): the whole loop in ONE cooperative launch, J in the CUs' LDS.
    # It is the product path for a problem of this size; the launch chain timed above stays on the line as `launch_chain`.
    chain = {k: out[k] for k in ("value", "ms_per_step")}
    chain.update({k: out["config"][k] for k in ("iterations_per_solve", "passes_per_solve", "rounds_per_solve", "us_per_round", "status",
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
            rp.upload_point(g["x0"], g["lower"], g["upper"])       # x, lower, upper: 384 bytes, as the launch chain uploads them per solve
            rp.launch()
            rp.stream.synchronize()
            steps_ms.append((time.perf_counter() - ts) * 1e3)
        rdt = time.perf_counter() - t0
        rres, rx, rst = rp.solve(g["x0"], g["lower"], g["upper"])  # the same solve once more for its result and in-kernel stamps
        riters = rres.iterations * K
        tick = 1e-2                                               # stats are in 10 ns ticks -> us
        rounds_r, passes_r = rst["rounds"], rst["passes"]
        out.update({"value": riters / rdt, "ms_per_step": rdt / K * 1e3})
        out["config"].update({
            "workload": f"cfg2 Gaussian-sum fit m={m} x n={n} fp64, width bounds, FD Jacobian; resident-J path: the whole LM loop in one "
                        "cooperative launch, J / y / row data in the CUs' LDS, compile-time residual model, default settings, whole solves",
            "path": "resident", "iterations_per_solve": rres.iterations, "passes_per_solve": passes_r, "rounds_per_solve": rounds_r,
            "fcalls_per_solve": rres.fCalls, "status": rres.status.name, "residual": rres.residual,
            "qp_active_set_passes_per_solve": rst["qp_active_set_passes"],
            "us_per_round": rdt / K / max(1, rounds_r) * 1e6, "us_per_pass": rdt / K / max(1, passes_r) * 1e6,
            "step_ms_min_median_max": [float(np.min(steps_ms)), float(np.median(steps_ms)), float(np.max(steps_ms))],
            "grid": rst["grid"], "rows_per_workgroup": rst["rows"], "lds_bytes_per_workgroup": rp.lds_bytes,
            "kernel_us_per_solve": rst["t_total"] * tick,
            "time_split_us_per_solve": {"workers_trial_residuals": rst["t_w_eval"] * tick, "workers_fd_refreshes": rst["t_w_fd"] * tick,
                                        "workers_products_and_publication": rst["t_w_prod"] * tick,
                                        "group_leaders": rst["t_group"] * tick, "wait_for_totals": rst["t_total_wait"] * tick,
                                        "solver_workgroup": rst["t_solver"] * tick, "of_which_n_x_n_solves": rst["t_solve_body"] * tick,
                                        "wait_for_command": rst["t_cmd_wait"] * tick, "staging": rst["t_stage"] * tick,
                                        "note": "stamps of workgroup 0 (s_memrealtime) inside the one launch"},
            "jacobian_full": rst["jacobian_full"], "jacobian_broyden": rst["jacobian_broyden"], "rejected": rst["rejected"],
            "elided_null_steps": rst["elided_evaluations"], "rejections_decided_by_lookahead": rst["lookahead_rejections"]})
        out["config"].pop("time_split_ms_per_solve", None)
        out["launch_chain"] = chain
        # HBM bytes of one launch from the committed rocprofv3 --pmc passes of this command (FETCH_SIZE doubled: the gfx950 correction)
        traffic, traffic_src, valu_busy = None, None, None
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cfg2_pmc.json")), key=_round_no):
            try:
                pm = next(v for k, v in json.load(open(f))["kernels"].items() if "k_lm_resident" in k)
                traffic, traffic_src, valu_busy = pm.get("hbm_bytes_per_launch"), os.path.relpath(f, ROOT), pm.get("valu_util")
            except (StopIteration, KeyError, ValueError):
                pass
        out["roofline"] = {"kernel": "mirlsq::k_lm_resident<ResGaussSum<5>, true> (the one launch of a solve)", "bound": "latency", "achieved": None,
                           "peak": None, "unit": "us", "frac": None, "avg_launch_ms": rst["t_total"] * tick / 1e3, "launches": 1, "traffic": traffic,
                           "traffic_source": traffic_src, "valu_busy_pmc": valu_busy,
                           "algorithmic_bytes_per_launch": float(m * 2 * 8 + 3 * n * 8),
                           "note": "J never leaves LDS: 16 MB of operands against 40 MB of LDS on the chip; a pass is three in-launch hand-offs "
                                   "(members -> 16 leaders -> workgroup 0 -> everybody) and a one-wave n = 16 solve -- latency, not HBM or MFMA. "
                                   "Figures of merit: us_per_pass, us_per_round"}
        res, x = rres, rx
    else:
        out["roofline"] = chain["solve_kernel"]
        out["config"]["path"] = f"launch chain (resident plan returned {rp.plan_rc})"
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        ctx = O.GaussSumCtx(g["t"].ctypes.data, g["data"].ctypes.data)
        t1 = time.perf_counter()
        ro, xo = O.optimize(O.native_fn("wlc_gauss_sum_f"), m, g["x0"], lower=g["lower"], upper=g["upper"], fctx=C.addressof(ctx))
        dtc = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": ro.iterations / dtc, "unit": "iterations/s", "cores": int(os.environ.get("OMP_NUM_THREADS", "1")),
                               "host_nproc": os.cpu_count(), "kind": "port",
                               "sample": f"the whole solve ({ro.iterations} iterations, fCalls {ro.fCalls}, status {O.STATUS.get(ro.status)}), {dtc:.2f} s, "
                                         "plain-loop BLAS, OpenMP residuals",
                               "parity_x_max_abs_diff": float(np.abs(np.asarray(x) - np.asarray(xo)).max()),
                               "parity_residual_rel_diff": abs(res.residual - ro.residual) / abs(ro.residual)}
    api.lib().mir_lsq_workspace_destroy(ws)
    print(json.dumps(out), flush=True)


def cfg5_roofline(ms, evals, steps, count, m, n):
    """k_lm_batched is neither HBM- nor MFMA-bound (8.4 MB of inputs per launch): its bound is the VALU issue rate -- one wave64
    instruction per 4 cycles per SIMD, 1024 SIMDs at 2.4 GHz = 614.4 G wave-instructions/s. `achieved` = the VALU instructions
    one launch executes (SQ_INSTS_VALU of the committed rocprofv3 pass, profiles/r03/cfg5_pmc.json: the instruction count of a
    launch does not depend on the box) over this run's launch time; `valu_busy_pmc` is the hardware's own figure
    (4 x SQ_ACTIVE_INST_VALU over GRBM_GUI_ACTIVE x 1024 SIMDs) from the same pass."""
    import glob
    peak = 1024 * 2.4e9 / 4.0 / 1e9
    pm, stale = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cfg5_pmc.json")), key=_round_no):
        try:
            doc = json.load(open(f))
            pm = next(v for k, v in doc["kernels"].items() if "k_lm_batched" in k)
            src = os.path.relpath(f, ROOT)
            # the instruction count of a launch belongs to the kernel source (and the LM settings) it was counted on: the
            # summary records the hash of batched_kernel.h; a different (or missing) hash leaves achieved / frac empty
            have, want = (doc.get("csrc_sha16") or {}).get("batched_kernel.h"), csrc_sha16("batched_kernel.h")
            stale = None if have == want else f"{src} was counted on batched_kernel.h {have}, this tree has {want}: re-profile (scripts/profile_any.sh cfg5 ... VALU SQ1)"
        except (StopIteration, KeyError, ValueError):
            pass
    out = {"kernel": "mirlsq::k_lm_batched<2> (one wavefront = one workgroup per problem: J, y in its 20 KB of LDS, FD + Broyden + "
                     "J^T J + posvx (one matrix row per lane) + acceptance in registers; no barrier, no host round trip)",
           "bound": "valu", "achieved": None, "peak": peak, "unit": "G wave64 VALU instructions/s", "frac": None,
           "avg_launch_ms": ms, "launches": steps, "traffic": None,
           "algorithmic_bytes_per_launch": float(count * (m * 4 + 2 * n * 4 + 24) + m * 4),
           "model_evaluations_per_s_upper_bound": evals / (ms * 1e-3),
           "note": "neither HBM- nor MFMA-bound: 8.4 MB of inputs per launch (< 1 % of the launch time at HBM rate). LDS allows two "
                   "waves per SIMD (20 KB a problem); while two are resident the VALU pipe is ~85 % busy, but the launch ends with "
                   "its longest fits (29 iterations where the mean is 10; 4096 problems on 2048 slots): on average 1.1 waves are "
                   "resident per SIMD (mean_resident_waves_per_simd), which is what holds the fraction near one half"}
    if stale:
        out["counters_stale"] = stale
    elif pm and pm.get("SQ_INSTS_VALU"):
        out["achieved"] = pm["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9
        out["frac"] = out["achieved"] / peak
        out["valu_instructions_per_launch"] = pm["SQ_INSTS_VALU"]
        out["transcendental_instructions_per_launch"] = pm.get("SQ_INSTS_VALU_TRANS_F32")
        out["valu_busy_pmc"] = pm.get("valu_util")
        if pm.get("SQ_WAVE_CYCLES") and pm.get("GRBM_GUI_ACTIVE"):      # SQ_WAVE_CYCLES counts 4-cycle units, GUI_ACTIVE sums 8 XCDs
            out["mean_resident_waves_per_simd"] = 4.0 * pm["SQ_WAVE_CYCLES"] / (pm["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        out["counters_source"] = src + " (committed rocprofv3 --pmc passes of this command; not measured in this run)"
    return out


def describe_comm(api, comm):
    if not comm:
        return None
    buf = C.create_string_buffer(512)
    api.lib().mir_lsq_comm_describe(comm, buf, 512)
    return buf.value.decode()


def step_stats(ms):
    import statistics
    return [min(ms), statistics.median(ms), max(ms)] if ms else None


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell (no launcher): this process becomes the PARENT of N rank processes
    -- `sys.executable bench.py <same arguments>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- and
    never touches torch or HIP itself. Rank 0's stdout is relayed (its JSON line is the parent's last stdout line), the
    other ranks' stdout goes to stderr. A rank that fails takes the job down: the others are terminated (by PID), the
    parent exits with that rank's code; nothing is retried. No os.exec* anywhere."""
    import socket
    import subprocess
    import threading

    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        # The image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (its host driver only supports dmabuf IPC: without it RCCL's P2P set-up
        # fails with `hipIpcGetMemHandle: invalid argument`). Whatever the box has set is INHERITED, never overridden; the default
        # is only supplied when the variable is missing altogether (a shell that lost the image's profile). DESIGN.md section 6.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(os.cpu_count() or 1, 64) // n)))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, cwd=os.getcwd()))
    lines = []

    def pump():
        for raw in procs[0].stdout:
            lines.append(raw.decode(errors="replace").rstrip("\n"))
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    failed = None
    deadline = time.monotonic() + float(os.environ.get("BENCH_RANK_TIMEOUT_S", "1500"))   # a rank stuck in a rendezvous must not hang the job
    while True:
        if time.monotonic() > deadline:
            failed = (-1, 124)
            print("[bench] ranks still running at the deadline (BENCH_RANK_TIMEOUT_S): stopping them", file=sys.stderr, flush=True)
            break
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    if failed is not None:
        if failed[0] >= 0:
            print(f"[bench] rank {failed[0]} exited with code {failed[1]}: stopping the other ranks", file=sys.stderr, flush=True)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    t.join(timeout=10)
    js = [l for l in lines if l.startswith('{"metric"')]
    for l in lines:                                  # everything rank 0 printed that is not the line goes to stderr
        if not l.startswith('{"metric"'):
            print(l, file=sys.stderr)
    sys.stderr.flush()
    if failed is not None:
        raise SystemExit(failed[1] if isinstance(failed[1], int) and 0 < failed[1] < 256 else 1)
    if not js:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr, flush=True)
        raise SystemExit(1)
    print(js[-1], flush=True)


def main():
    args = parse()
    if args.config == "cfg5":
        return main_cfg5(args)
    if args.config == "cfg2":
        os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
        return main_cfg2(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)                    # before torch / HIP are imported: the parent never initialises the GPU
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))   # CPU baseline leg (oracle, OpenMP)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (self-launching) or "
                         "torch.distributed.run --nproc-per-node N")
    import numpy as np
    import torch
    import torch.distributed as dist

    import mir_optim_amd as M
    from mir_optim_amd import api, parallel as PAR, workloads as W

    if not torch.cuda.is_available() or M.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: mir_optim_amd has no CPU path")
    # rehearsals: ranks share the GPUs there are (asked for, or forced: fewer visible devices than ranks -- RCCL then refuses
    # the communicator and the run takes the labelled callback fallback below instead of dying in set_device)
    ndev = torch.cuda.device_count()
    share = args.comm == "gloo-callback" or os.environ.get("BENCH_SHARE_GPU") == "1" or ndev < world
    if ndev < world and rank == 0:
        print(f"[bench] {world} ranks on {ndev} visible GPU(s): ranks share devices (rehearsal, not a scaling measurement)",
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
        # Control plane (unique-id exchange, the barriers around the timed region, the max over ranks): torch.distributed.
        # Data plane (every collective of the solve): the solver's OWN RCCL communicator over xGMI, created below.
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
            # the solver's own RCCL communicator (xGMI), id via torch.distributed; checked with one all-reduce of a known
            # payload before anything is timed. If ANY rank fails to create or verify it, every rank falls back to the callback
            # communicator over the control plane -- slower, labelled in config, but a measured line instead of a crash.
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
                print(f"[bench] rank {rank}: RCCL communicator unusable ({err or 'on another rank'}); falling back to the callback "
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
        tape, rres, rx, rwall = PAR.record_rank_tape(shard, replay, data["x0"], settings=settings, batched=fdb, variant=args.variant)
        inner = comm                                            # --force-comm: the one-rank RCCL communicator created above
        inner_comm = inner
        comm = PAR.replay_comm(replay, 0, tape, inner)
        replay_info = {"ranks": replay, "tape_doubles": int(tape.size), "grouped_solve_wall_ms": rwall * 1e3,
                       "grouped_solve": {"status": rres.status.name, "iterations": rres.iterations, "fcalls": rres.fCalls,
                                         "residual": rres.residual},
                       "inner": "one-rank RCCL all-reduce behind every replayed exchange" if inner else None,
                       "note": "value = iterations of the GLOBAL solve per second as ONE rank would deliver them with a zero-latency "
                               "interconnect: rank 0's shard on an otherwise idle GPU, every all-reduce replaced by the recorded total "
                               "of the real " + str(replay) + "-shard run (stream-ordered device copy)"}
        args.survey_steps = 0                                   # the tape belongs to the headline settings

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def solve(stats=None, flags=0, s=settings):
        if replay:
            api.lib().mir_lsq_comm_replay_rewind(comm)
        return prob.solve(data["x0"], settings=s, stats=stats, flags=flags, comm=comm_obj or comm, workspace=ws, variant=args.variant,
                          batched=fdb)

    # RCCL finishes part of its initialisation asynchronously: a few seconds after ncclCommInitRank every HIP launch of the
    # process stalls once or twice for 60-150 ms (measured on a one-GPU box with --force-comm: 160-310 it/s when that lands
    # in the timed region, 740-760 when not; RCCL / NCCL knobs and warm collectives do not move it). Instead of sleeping a
    # fixed time, run untimed solves and WATCH for it: a solve that takes more than 4x the fastest one seen is the stall;
    # the loop ends once a stall has been seen and 20 solves in a row are back to normal, or at --stall-bound seconds after
    # communicator creation. All ranks take the same decision (the flag is max-reduced over the control plane).
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
        # two statistics records: `st` for the steps whose kernels are bracketed with HIP events (its per-launch figures --
        # milliseconds, pending columns, points per call -- all refer to the same launches), `st_all` for every step (counters)
        st, st_plain = M.Stats(), M.Stats()
        iters = 0
        every = max(1, args.timing_every)
        step_ms, step_timed = [], []                         # host wall time of every step (a solve ends with a host wait)
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
        d["all"] = {k: (d[k] + dp[k]) if not isinstance(d[k], list) else [a + b for a, b in zip(d[k], dp[k])] for k in d}
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
                  # which of the two branches the noise-decided last acceptance took
                  "branch": ("xConverged after the confirming step (short: ~12-22 passes)" if r9.status.name == "xConverged"
                             else "the confirming step was rejected: the reference's lambda ladder runs to maxLambda (~45 more rejected passes)"),
                  "ms_per_step_min_median_max": step_stats(timed.last_steps[0])}

    out = None
    if rank == 0:
        K = args.steps
        value = iters / dt                                   # iterations of the GLOBAL solve per second, at every N
        nb = max(1, st["jtj_broyden_launches"])
        kern_ms = st["jtj_broyden_ms"] / nb
        survey_bytes = 8.0 * (2.0 * m * n + 3.0 * m)         # T (2 m n + 3 m), Broyden pass with J rewritten
        # broyden_lr.h: J is read once and never written; the sweep also reads the k pending columns of U, y_new, y_old and
        # writes one column: T (m n + (k + 3) m), k averaged over the timed launches
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
            fd_name = (f"mirlsq::k_jtj_fdp<{ncb}, false, true>" if diff_panel else f"mirlsq::k_jtj_fdp<{ncb}, true, false>") if n <= 128 \
                else f"mirlsq::k_jtj_fdp8<{ncb}, {'true' if diff_panel else 'false'}>"
            # read the panel (m x n differences, or m x 2n pairs) and y, write J
            fd_bytes = 8.0 * ((2.0 if diff_panel else 3.0) * m * n + m)
            fd_rate = fd_bytes / (fd_ms * 1e-3) / 1e9
            fd_tf = (jtj_flops + 2.0 * m * n) / (fd_ms * 1e-3) / 1e12
            # which roofline bounds it: n (n + 1) flop against 16 (or 24) bytes per row element -- at n = 128 the HBM time at
            # 8 TB/s (0.26 ms) exceeds the MFMA time at 78.6 TF (0.21 ms), at n = 256 it is the other way round (0.51 vs 0.84 ms)
            mfma_bound = (jtj_flops / (F64_MFMA_PEAK_TF * 1e12)) > (fd_bytes / (HBM_PEAK_GBS * 1e9))
            fresh = {
                "kernel": fd_name + (" (finite-difference rows from the m x n DIFFERENCE panel" if diff_panel else
                                     " (finite-difference rows from the (+h, -h) pair panel")
                                  + " -> J, J^T J + J^T y on f64 MFMA 16x16x4, register-staged producer waves + MFMA consumer waves)",
                **({"bound": "mfma", "achieved": fd_tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": fd_tf / F64_MFMA_PEAK_TF,
                    "hbm_GBs": fd_rate, "hbm_frac": fd_rate / HBM_PEAK_GBS} if mfma_bound else
                   {"bound": "hbm", "achieved": fd_rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fd_rate / HBM_PEAK_GBS}),
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
                "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TF,
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
        if st["fd_callback_calls"] and st["fd_callback_ms"] > 0:    # (two-stream window refreshes overlap the caller's kernels with the library's: not timed apart)
            ms = st["fd_callback_ms"] / st["fd_callback_calls"]
            pts = st["fd_callback_points"] / st["fd_callback_calls"]
            if args.fd == "serial":
                by = pts * 8.0 * (m * n + m)
                rate = by / (ms * 1e-3) / 1e9
                user["residual_gemm"] = {"kernel": "wl k_tanh_linear (one sweep over A per finite-difference point)", "bound": "hbm",
                                         "achieved": rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rate / HBM_PEAK_GBS,
                                         "avg_call_ms": ms, "calls": st["fd_callback_calls"], "points_per_call": pts,
                                         "algorithmic_bytes_per_call": by}
            else:
                fl = 2.0 * m * n * pts                          # A[m x n] . X^T[n x p]; + one tanh per output
                tf = fl / (ms * 1e-3) / 1e12
                diff_panel = args.fd == "batched" and ((n <= 128 and n % 2 == 0) or n in (192, 256))
                by = 8.0 * (m * n + m * pts * (0.5 if diff_panel else 1.0) + m)   # read A once, write the panel
                kn = "k_tanh_linear_batched_dma"
                kn_full = f"k_tanh_linear_batched_dma<{n // 4}, true, {'true' if diff_panel else 'false'}>"
                user["residual_gemm"] = {"kernel": f"wl {kn} (caller side: the 2n finite-difference points as one A . X^T GEMM on f64 MFMA "
                                                   "+ tanh epilogue, writes the " + ("m x n difference panel)" if diff_panel else "m x 2n panel)"),
                                         "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                         "frac": tf / F64_MFMA_PEAK_TF, "avg_call_ms": ms, "calls": st["fd_callback_calls"],
                                         "points_per_call": pts, "algorithmic_bytes_per_call": by,
                                         "algorithmic_GBs": by / (ms * 1e-3) / 1e9,
                                         "traffic": pmc_field(kn_full, m, n, "hbm_bytes_per_launch"),
                                         "mfma_util_pmc": pmc_field(kn_full, m, n, "mfma_util"),
                                         "traffic_source": traffic_source(m, n)}
        if st["trial_callback_calls"]:
            ms = st["trial_callback_ms"] / st["trial_callback_calls"]
            pts = st["trial_callback_points"] / st["trial_callback_calls"]
            by = 8.0 * (m * n + pts * m + m)                    # one sweep over A serves the points of a call (ladder trials)
            rate = by / (ms * 1e-3) / 1e9
            user["trial_residual"] = {"kernel": "wl k_tanh_linear / k_tanh_linear_multi (caller side: f(trial), one sweep over A per call)",
                                      "bound": "hbm", "achieved": rate, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": rate / HBM_PEAK_GBS, "avg_call_ms": ms, "calls": st["trial_callback_calls"],
                                      "points_per_call": pts, "algorithmic_bytes_per_call": by}
        solve_k = {"kernel": "mirlsq::k_lm_solve (damping, posvx('E','L'), BOXCQP, step rounding, prediction: one workgroup per ladder entry)",
                   "bound": "latency", "avg_launch_ms": st["solve_ms"] / max(1, st["solve_launches"]),
                   "launches": st["solve_launches"], "flops_per_launch": n ** 3 / 3.0}
        KT = max(1, timed_steps)                             # steps of the timed region whose kernels were event-timed
        lib_ms = (st["jtj_ms"] + st["solve_ms"]) / KT
        user_ms = (st["fd_callback_ms"] + st["trial_callback_ms"]) / KT
        out = {
            "metric": "LM iterations/sec", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
            # host wall time per step on rank 0 (min, median, max): all K steps, and the steps without kernel events only
            "ms_per_step_min_median_max": step_stats(main_steps[0]),
            "ms_per_step_uninstrumented_min_median_max": step_stats([t for t, e in zip(*main_steps) if not e]),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"cfg3 tanh-linear NLS m_total={m_total} x n={n} fp64, FD Jacobian (central differences, 2n residual evaluations per refresh through the "
                            f"{ {'batched': 'batched difference-panel', 'rowmajor': 'batched pair-panel', 'pointmajor': 'batched point-major', 'serial': 'single-point'}[args.fd] } callback), "
                            f"absTolerance={args.abs_tolerance:g}, whole solves x0 -> termination, {args.scaling} scaling "
                            f"({m} rows on rank 0)",
                "m_total": m_total, "m_per_gpu": m, "n": n, "scaling": args.scaling,
                "parallelism": (f"ONE rank of {replay} (rows sharded x{replay}; all-reduce totals replayed from the recorded {replay}-shard solve)"
                                if replay else f"rows sharded x{world}, ") + ("" if replay else "RCCL all-reduce" if args.comm == "rccl" else
                                                                 "gloo callback all-reduce (" + ("FALLBACK: RCCL unusable" if comm_fallback else "rehearsal") + ")"),
                "rccl_ranks": api.lib().mir_lsq_comm_ranks(comm) if (comm and args.comm == "rccl") else None,
                "rccl_fallback_reason": comm_fallback,
                "launcher": ("bench.py (self-launched rank processes)" if os.environ.get("BENCH_SELF_LAUNCHED") == "1"
                             else "external (torch.distributed.run)") if world > 1 else None,
                "ranks_share_gpus": bool(share and world > 1), "visible_gpus": ndev,
                "comm": describe_comm(api, comm),     # transport, the shared object RCCL was bound from, its version, ncclCommCount
                # library kernel launches per round, by the kind of round (refresh / Broyden / re-solve after a rejection)
                "library_launches_per_round": {k: (sta["round_launches"][i] / sta["rounds"][i] if sta["rounds"][i] else None)
                                               for i, k in enumerate(("refresh", "broyden", "resolve"))},
                "rounds_per_solve": {k: sta["rounds"][i] / K for i, k in enumerate(("refresh", "broyden", "resolve"))},
                "allreduce_per_solve": {"packed_calls": sta["allreduce_calls"][0] / K, "packed_elems": sta["allreduce_elems"][0] / max(1, sta["allreduce_calls"][0]),
                                        "sweep_calls": sta["allreduce_calls"][1] / K, "sweep_elems": sta["allreduce_elems"][1] / max(1, sta["allreduce_calls"][1]),
                                        "scalar_calls": sta["allreduce_calls"][2] / K},
                "rccl_stall_probe": stall if t_comm is not None else None,
                "replay": replay_info,
                "abs_tolerance": args.abs_tolerance,
                "abs_tolerance_note": "1e-5: every accept/reject decision of the solve has margin; at the survey's 1e-9 the last "
                                      "acceptance compares rounding noise (12 it / 16 passes or 11 it / 56 passes): see survey_setting",
                "survey_setting": survey,
                "iterations_per_solve": iters / K, "status": res.status.name,
                "passes_per_solve": sta["passes"] / K, "fcalls_per_solve": res.fCalls,
                "jacobian_full_per_solve": sta["jacobian_full"] / K, "residual": res.residual,
                "kernel_timing": f"HIP events on the solver's stream in {timed_steps} of the {K} timed steps (every {max(1, args.timing_every)}th)",
                "time_split_ms_per_solve": {
                    "caller_fd_callbacks": st["fd_callback_ms"] / KT, "caller_trial_callbacks": st["trial_callback_ms"] / KT,
                    "jtj_fd_kernel": st["jtj_fd_ms"] / KT, "broyden_sweep": st["jtj_broyden_ms"] / KT,
                    "solve_kernel": st["solve_ms"] / KT, "library_kernels": lib_ms, "caller_kernels": user_ms,
                    "total_wall": sta["total_ms"] / K},
            },
            # `roofline` = the kernel with the most time in the TIMED REGION, caller-side kernels included (round-4 review: the
            # caller's GEMM is 46 % of the GPU time at cfg 3, the library's busiest kernel 13 %); the two hot library kernels always
            # have their own objects (jtj_kernel, broyden_kernel), the caller's theirs (residual_gemm, trial_residual)
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
                               avg_launch_ms=obj.get("avg_launch_ms", obj.get("avg_call_ms")), launches=obj.get("launches", obj.get("calls")),
                               library_dominant={"object": "jtj_kernel" if fresh_total >= sweep_total else "broyden_kernel",
                                                 "frac": dominant["frac"], "bound": dominant["bound"]})
        if world == 1 and not args.no_host_callback and (m, n) == (1_000_000, 128):
            # the path a caller of the UNMODIFIED reference API gets: host residual callback, native thread manager, PCIe inclusive
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import bench_host_callback as BH
                hc = BH.run(m, n, abs_tolerance=args.abs_tolerance, data=data, solves=1)
                xh = hc.pop("x")
                hc["parity_x_max_abs_diff_vs_device_callback_solve"] = float(np.abs(np.asarray(xh) - np.asarray(x)).max())
                out["host_callback_mode"] = hc
            except Exception as e:      # noqa: BLE001 -- an auxiliary leg must not take the headline line down with it
                out["host_callback_mode"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, m, n, args.cpu_iterations, args.abs_tolerance, min(os.cpu_count() or 1, 64), x, res)
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
    return {**parity, "value": ro.iterations / dt, "unit": "iterations/s", "cores": threads, "host_nproc": os.cpu_count(),
            "kind": "port",
            "sample": f"the first {ro.iterations} accepted LM iteration(s) of the same m={m} x n={n} solve (bounded by "
                      f"maxIterations={iterations}; status {O.STATUS.get(ro.status, ro.status)}, "
                      f"fCalls {ro.fCalls}: FD Jacobians of {2 * n} residual calls each + Broyden passes), {dt:.1f} s, "
                      f"OpenBLAS={'yes' if ob else 'no (plain loops)'} x{threads}, residual calls OpenMP x{threads}",
            "seconds": dt, "fcalls": ro.fCalls}


if __name__ == "__main__":
    main()
