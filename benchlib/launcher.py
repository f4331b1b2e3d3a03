"""bench.py --gpus N from a bare shell: this process starts the N rank processes itself and relays rank 0's JSON line
(the parent never imports torch or touches HIP)."""
import ctypes as C
import json
import os
import sys
import time

from .common import (F64_MFMA_PEAK_TF, HBM_PEAK_GBS, ROOT, _round_no, csrc_sha16, describe_comm, flush_c_stdio,
                     # noqa: F401
                     pmc_field, pmc_file, step_stats, traffic_source)


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell (no launcher): this process becomes the PARENT of N rank processes
    -- `sys.executable bench.py <same arguments>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set --
    and
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
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        # The image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (its host driver only supports dmabuf IPC: without it RCCL's
        # P2P set-up fails with `hipIpcGetMemHandle: invalid argument`). Whatever the box has set is INHERITED, never
        # overridden; the default is only supplied when the variable is missing altogether (a shell that lost the
        # image's profile). DESIGN.md section 6.
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
    deadline = time.monotonic() + float(os.environ.get("BENCH_RANK_TIMEOUT_S",
                                                       "1500"))   # a rank stuck in a rendezvous must not hang the job
    while True:
        if time.monotonic() > deadline:
            failed = (-1, 124)
            print("[bench] ranks still running at the deadline (BENCH_RANK_TIMEOUT_S): stopping them", file=sys.stderr,
                  flush=True)
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
            print(f"[bench] rank {failed[0]} exited with code {failed[1]}: stopping the other ranks", file=sys.stderr,
                  flush=True)
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
