"""Row-shard data parallelism for the LM solver: rank r owns a contiguous block of
the m residual rows; x, dx, lambda, J^T J, J^T y and every scalar are replicated and every rank runs the same
control flow (the n x n solve is replicated too: its inputs are bit-identical after the all-reduce, so no
broadcast is needed). The only exchange is a sum all-reduce of the packed [J^T J lower | J^T y] buffer per
Jacobian-changing pass and of one scalar per residual evaluation.

Two communicators exist in the C library: RCCL (one process per GPU over xGMI; the production path) and a
callback communicator whose all-reduce is supplied by the host program -- used here with torch.distributed
`gloo` so that the sharded path can be exercised with world_size 2 on a single GPU or on CPU-only hosts."""
import ctypes as C

import numpy as np

from . import api


def row_shard(m_total, world, rank):
    """Contiguous, balanced row blocks: (row_offset, m_local). The first m_total % world ranks get one extra row."""
    base, extra = divmod(int(m_total), int(world))
    m_local = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, m_local


def packed_length(n):
    """Doubles in the fused all-reduce payload: n(n+1)/2 (J^T J lower) + n (J^T y)."""
    return n * (n + 1) // 2 + n


def rccl_comm(world, rank, broadcast_bytes):
    """Create the RCCL communicator. `broadcast_bytes(buf: np.ndarray[uint8, 128]) -> np.ndarray` must return rank
    0's buffer on every rank (any out-of-band channel: torch.distributed, MPI, a file ...)."""
    L = api.lib()
    uid = np.zeros(128, dtype=np.uint8)
    if rank == 0 and L.mir_lsq_rccl_unique_id(uid.ctypes.data) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    uid = np.ascontiguousarray(broadcast_bytes(uid), dtype=np.uint8)
    comm = L.mir_lsq_comm_create_rccl(world, rank, uid.ctypes.data)
    if not comm:
        raise RuntimeError("ncclCommInitRank failed")
    return comm


def check_comm(comm, world, rank, count=8384):
    """One all-reduce of a known payload through the communicator (the packed J^T J + J^T y length of n = 128 by default):
    rank r contributes (r + 1) (k % 7 + 1) at element k. True when every element comes back as the exact sum."""
    buf = np.arange(count, dtype=np.float64) % 7 + 1
    d = api.DeviceBuffer(buf * (rank + 1))
    if api.lib().mir_lsq_comm_allreduce_d(comm, d.ptr, count, None) != 0:
        return False
    got = d.download()
    d.free()
    return bool(np.array_equal(got, buf * (world * (world + 1) // 2)))


class HostAllreduceComm:
    """Callback communicator: device buffer -> host -> `allreduce_numpy(buf)` (in place, sum) -> device."""

    def __init__(self, world, rank, allreduce_numpy):
        self.allreduce_numpy = allreduce_numpy
        L = api.lib()

        self.errors = []     # exceptions inside the callback (ctypes would swallow them): see check()

        def cb(_ctx, dev_ptr, count, stream):
            host = np.empty(count, dtype=np.float64)
            try:
                if L.mir_lsq_memcpy_d2h(host.ctypes.data, dev_ptr, count * 8, stream) != 0:
                    raise RuntimeError("D2H failed in all-reduce callback")
                self.allreduce_numpy(host)
            except BaseException as e:      # noqa: BLE001 -- poison the payload: the solve ends with numericError
                self.errors.append(e)
                host[:] = np.nan
            if L.mir_lsq_memcpy_h2d(dev_ptr, host.ctypes.data, count * 8, stream) != 0:
                self.errors.append(RuntimeError("H2D failed in all-reduce callback"))

        self._cb = api.ALLREDUCE_FN(cb)
        self.handle = L.mir_lsq_comm_create_callback(world, rank, self._cb, None)
        if not self.handle:
            raise RuntimeError("mir_lsq_comm_create_callback failed")

    def check(self):
        """Re-raise the first exception a callback recorded (call after the solve returns)."""
        if self.errors:
            raise self.errors[0]

    def close(self):
        if self.handle:
            api.lib().mir_lsq_comm_destroy(self.handle)
            self.handle = None


def torch_allreduce_numpy(dist):
    """`allreduce_numpy` over a torch.distributed process group (gloo on CPU tensors)."""
    import torch

    def f(buf):
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return f


def local_group(world):
    """In-process group (mir_lsq_comm_create_local_group): `world` communicator handles for `world` solver instances
    in THIS process, one host thread each. Returns (handles, close)."""
    L = api.lib()
    arr = (C.c_void_p * world)()
    if L.mir_lsq_comm_create_local_group(world, arr) != 0:
        raise RuntimeError("mir_lsq_comm_create_local_group failed")
    handles = [arr[r] for r in range(world)]

    def close():
        for h in handles:
            L.mir_lsq_comm_destroy(h)
    return handles, close


def record_rank_tape(make_problem, world, x0, rank=0, capacity=1 << 22, **solve_kw):
    """Run the `world`-shard solve ONCE in this process (in-process group, one host thread per shard, all on the current
    device) and record the totals of rank `rank`'s all-reduces (mir_lsq_comm_record): the tape a replay communicator
    (replay_comm) needs to let that one rank re-run the GLOBAL trajectory alone. make_problem(r) builds shard r's
    DeviceProblem. Returns (tape: np.ndarray[float64], result of `rank`, x, wall seconds of the grouped solve)."""
    import threading
    import time

    L = api.lib()
    comms, close = local_group(world)
    probs = [make_problem(r) for r in range(world)]
    tape = np.zeros(capacity, dtype=np.float64)
    if L.mir_lsq_comm_record(comms[rank], tape.ctypes.data, capacity) != 0:
        raise RuntimeError("mir_lsq_comm_record failed")
    res, err = [None] * world, [None] * world

    def one(r):
        try:
            res[r] = probs[r].solve(x0, comm=comms[r], **solve_kw)
        except BaseException as e:   # noqa: BLE001
            err[r] = e
    t0 = time.perf_counter()
    ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(1200)
    wall = time.perf_counter() - t0
    if any(t.is_alive() for t in ts):
        # a solver thread is still inside an all-reduce on these handles: LEAK the group rather than free it under a live thread
        raise RuntimeError("grouped solve did not finish within the join timeout (the communicators are left allocated)")
    n = L.mir_lsq_comm_recorded(comms[rank])
    close()
    if any(err):
        raise RuntimeError(f"grouped solve failed: {err}")
    if n == C.c_size_t(-1).value:
        raise RuntimeError("tape overflow: raise `capacity`")
    return tape[:n].copy(), res[rank][0], res[rank][1], wall


def replay_comm(world, rank, tape, inner=None):
    """mir_lsq_comm_create_replay: every all-reduce is replaced by the next recorded total (and also passes through
    `inner`, e.g. a one-rank RCCL communicator, when given). Rewind it (mir_lsq_comm_replay_rewind) before every solve."""
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    h = api.lib().mir_lsq_comm_create_replay(world, rank, tape.ctypes.data, tape.size, inner)
    if not h:
        raise RuntimeError("mir_lsq_comm_create_replay failed")
    return h
