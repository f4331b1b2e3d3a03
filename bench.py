#!/usr/bin/env python3
"""bench.py -- LM iterations/sec on the BASELINE.json headline workload (the configurations live in benchlib/).

  python bench.py --gpus N --steps K --warmup W
  N > 1 from a bare shell: this process starts the N rank processes itself (benchlib/launcher.py: children of this
  interpreter with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; the parent never imports torch or touches HIP) and
  relays rank 0's JSON line. Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (WORLD_SIZE set) it is one rank.

Workload (cfg 3): tanh-linear synthetic NLS, r_i(x) = tanh(a_i . x) - b_i, m = 1e6 rows x n = 128
parameters, fp64, finite-difference Jacobian through the user's batched residual callbacks, defaults except absTolerance
(1e-5: every accept / reject decision of the solve has margin; the survey's 1e-9 is measured too and reported in
config.survey_setting -- there the last acceptance compares rounding noise).

Scaling (BASELINE.json: "m=1e6 x n=128 ... 1/2/4/8 MI355X"):
  --scaling strong (default)  the SAME 1e6-row problem, rows split over the N ranks (rank r owns rows
                              row_shard(1e6, N, r)); `value` = LM iterations of that one global solve per second.
  --scaling weak              1e6 rows PER GPU (cfg 4's partition; the global problem grows with N); `value` is still
                              the global solve's iterations per second -- it does NOT multiply by N.
Per solve the ranks exchange one sum all-reduce of the packed [J^T J | J^T y] per full refresh, ONE of the
[sweep vector | trial sum of squares] (2n + 35 doubles) per trial of a fused round, and one scalar for the entry
residual, on the solver's own RCCL communicator (cfg 3: 9 collectives per solve).

A "step" is one complete LM solve (mir_optimize_least_squares_gpu_d from x0 to termination): residual + FD Jacobian
callbacks, Broyden updates, J^T J / J^T y, damped BOXCQP solves, step acceptance -- nothing is skipped or cached between
solves. Inputs are resident in HBM before the timed region.

The JSON line also carries
  roofline        the kernel with the most time in the timed region -- caller-side kernels included (at cfg 3 it is the
                  caller's finite-difference GEMM) --, HIP-event timed on the solver's stream; `object` names the entry
                  it copies
  jtj_kernel, broyden_kernel     the two hot LIBRARY kernels (fused FD / plain J^T J; the Broyden sweep)
  residual_gemm, trial_residual  the CALLER-side device callbacks (the synthetic workload's kernels, csrc/workloads.hip
                  + workloads_gemm.hip), event-timed on the same stream: they are most of a solve and get their own
                  roofline objects
  solve_kernel    the one-workgroup n x n kernel (latency-bound; time only)
  cpu_baseline    the oracle (CPU port of the reference algorithm, OpenBLAS for syrk/gemv/ger/posvx) on a bounded sample
                  of the same workload at min(nproc, 64) threads, plus cpu_baseline_1thread; rank 0, N = 1 only.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (the host driver only supports dmabuf IPC; without it RCCL's peer setup
# fails with hipIpcGetMemHandle: invalid argument). A launcher that builds its own environment may drop it: keep it, before
# anything loads the HIP runtime.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["cfg3", "cfg2", "cfg5"], default="cfg3",
                    help="cfg3: the headline (BASELINE.json metric). cfg2: BASELINE config 2, Gaussian-sum fit m = 1e5 "
                         "x n = 16 fp64 "
                         "(launch-latency bound). cfg5: BASELINE config 5, 4096 independent fp32 fits of m = 512 x n = "
                         "8, one "
                         "wavefront per problem. cfg2 / cfg5 print additional lines (one GPU; replicas only at N > 1)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    # (under torch.distributed.run, abbreviations such as --m collide with the launcher's own options: BENCH_M /
    # BENCH_N)
    ap.add_argument("--rows", type=int, default=int(os.environ.get("BENCH_M", 1_000_000)),
                    help="rows of the global problem (strong) or per GPU (weak)")
    ap.add_argument("--n", type=int, default=int(os.environ.get("BENCH_N", 128)))
    ap.add_argument("--scaling", choices=["strong", "weak"], default=os.environ.get("BENCH_SCALING", "strong"))
    ap.add_argument("--fd", choices=["batched", "rowmajor", "pointmajor", "serial"], default="batched",
                    help="finite differences through the batched residual callbacks -- batched: the m x n row-major "
                         "DIFFERENCE "
                         "panel (n <= 128; the caller's kernel subtracts the (+h, -h) pair, the library's fused kernel "
                         "scales, writes J "
                         "and forms J^T J); rowmajor: the m x 2n row-major pair panel, same fused kernel; pointmajor: "
                         "the point-major "
                         "batched callback + k_fd_fill -- or one call per point (serial)")
    ap.add_argument("--gemm-read-a-once", action="store_true",
                    help="caller side, --fd batched: the difference-panel GEMM sweeps A once (stage-outer variant: 2.1 "
                         "instead of "
                         "3.1 GB per call at cfg 3, ~3 %% slower -- the kernel is MFMA-bound; A/B only)")
    ap.add_argument("--abs-tolerance", type=float, default=1e-5,
                    help="LeastSquaresSettings.absTolerance of the headline number (DESIGN.md section 5)")
    ap.add_argument("--survey-steps", type=int, default=10,
                    help="solves timed at SURVEY 8d's absTolerance = 1e-9 after the main region (0 = skip)")
    ap.add_argument("--control-plane", choices=["gloo", "nccl"], default="gloo",
                    help="torch.distributed backend for barriers / id exchange (the solve's collectives always use the "
                         "solver's own RCCL communicator)")
    ap.add_argument("--comm", choices=["rccl", "gloo-callback"], default="rccl",
                    help="data-plane communicator: the solver's RCCL communicator (production) or, to rehearse N > 1 "
                         "on a box "
                         "with fewer GPUs, its callback communicator over gloo (ranks then share GPUs)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N = 1: attach a one-rank RCCL communicator anyway (exercises the N > 1 code path on one GPU)")
    ap.add_argument("--replay-ranks", type=int, default=0,
                    help="N = 1 only: measure ONE rank of an R-rank strong-scaled job on this GPU. The R-shard solve "
                         "of the global "
                         "problem runs once (in-process group) and records rank 0's all-reduce totals; every timed "
                         "solve is then "
                         "rank 0's shard alone, each exchange replaced by the recorded total "
                         "(mir_lsq_comm_create_replay) -- the "
                         "global trajectory, the rank's own kernels uncontended, no xGMI hop. With --force-comm every "
                         "exchange "
                         "also passes through a one-rank ncclAllReduce")
    ap.add_argument("--replay-latency-us", type=int, default=0,
                    help="with --replay-ranks: a MODEL of the collectives' latency -- every replayed exchange first "
                         "holds the stream this many microseconds (mir_lsq_comm_replay_set_delay); the line says so")
    ap.add_argument("--stall-bound", type=float, default=8.0,
                    help="RCCL only: upper bound (s, from communicator creation) of the warm-up loop that waits for "
                         "RCCL's "
                         "asynchronous initialisation stall to pass (it ends as soon as a stall has been seen and has "
                         "passed)")
    ap.add_argument("--variant", type=int, default=0, help="MIR_LSQ_VARIANT_* bits for A/B runs (0 = product path)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region (A/B of the instrumentation "
                         "overhead; the "
                         "roofline objects are then empty)")
    ap.add_argument("--timing-every", type=int, default=10,
                    help="bracket the kernels with HIP events in every k-th step of the timed region (an event record "
                         "costs a "
                         "few microseconds on the stream: ~0.2 ms per cfg-3 solve when every step is instrumented)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cfg5-replicas", type=int, default=16,
                    help="--config cfg5: also time ONE launch of this many copies of the 4096 problems (the "
                         "steady-state rate of the kernel, "
                         "where the tail of long fits is amortised); 1 = skip")
    ap.add_argument("--no-host-callback", action="store_true",
                    help="skip the reference-ABI leg (host residual callback + native thread manager, PCIe inclusive; "
                         "rank 0, N = 1): "
                         "one untimed + one timed solve, a few seconds of host work")
    ap.add_argument("--no-cpu-1thread", action="store_true")
    ap.add_argument("--cpu-iterations", type=int, default=6, help="accepted iterations of the CPU sample")
    return ap.parse_args()


def main():
    args = parse()
    if args.config == "cfg5":
        from benchlib.cfg5 import main_cfg5
        return main_cfg5(args)
    if args.config == "cfg2":
        os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
        from benchlib.cfg2 import main_cfg2
        return main_cfg2(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from benchlib.launcher import launch_ranks
        # before torch / HIP are imported: the parent never initialises the GPU
        return launch_ranks(args)
    from benchlib.cfg3 import main_cfg3
    return main_cfg3(args)


if __name__ == "__main__":
    main()
