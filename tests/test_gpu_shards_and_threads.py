"""Row sharding with R logical shards on ONE GPU and re-entrancy (8b "Threading").

The GPU box allows six processes on the card, so eight shards cannot be eight processes: they are eight HOST THREADS of
this process, each with its own solver instance, stream, workspace, data shard and a handle of the library's in-process
communicator group (mir_lsq_comm_create_local_group: device -> pinned host, barrier, sum in rank order, host -> device).
ctypes releases the GIL for the duration of the C call and the callbacks are native device callbacks, so the eight solves
really run concurrently -- which makes every one of these tests a re-entrancy test of the library as well.

Reference: the three reductions of a pass are least_squares.d:1052 (J^T y), 1065 (J^T J) and 1115 (||f(trial)||^2)."""
import ctypes as C
import threading

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import parallel as PAR
from mir_optim_amd import workloads as W
import problems as P

pytestmark = pytest.mark.gpu


def run_threads(fns):
    out, err = [None] * len(fns), [None] * len(fns)

    def wrap(i):
        try:
            out[i] = fns[i]()
        except BaseException as e:     # noqa: BLE001 -- reported below, in the main thread
            err[i] = e
    ts = [threading.Thread(target=wrap, args=(i,)) for i in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not any(t.is_alive() for t in ts), "a shard thread hangs"
    for e in err:
        if e is not None:
            raise e
    return out


def sharded_solve(m_total, n, world, settings, lower=None, upper=None, x0=None, **kw):
    comms, close = PAR.local_group(world)
    probs, stats = [], []
    for r in range(world):
        off, ml = PAR.row_shard(m_total, world, r)
        w = P.tanh_linear(ml, n, row_offset=off)
        probs.append((W.TanhLinear(w["A"], w["b"]), w))
        stats.append(M.Stats())
    start = probs[0][1]["x0"] if x0 is None else x0

    def one(r):
        prob, w = probs[r]
        return lambda: prob.solve(start, l=lower, u=upper, settings=settings, comm=comms[r], stats=stats[r], batched=True, **kw)
    try:
        res = run_threads([one(r) for r in range(world)])
    finally:
        close()
    return res, stats


def oracle_solve(oracle, m_total, n, settings_fn, lower=None, upper=None, x0=None):
    w = P.tanh_linear(m_total, n)
    so = oracle.default_settings()
    settings_fn(so)
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    return oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), m_total, w["x0"] if x0 is None else x0, lower=lower, upper=upper,
                           settings=so, fctx=C.addressof(ctx), use_openblas=oracle.load_openblas(threads=8))


@pytest.mark.parametrize("m_total,n,world", [(80000, 128, 8), (40004, 256, 8), (30001, 64, 3), (9000, 33, 8)])
def test_eight_logical_shards_equal_the_unsharded_oracle(oracle, m_total, n, world):
    """cfg 3's n = 128 and cfg 4's n = 256 (rows split 8 ways, m scaled down), plus ragged shards and an odd n:
    every rank returns the same bits; x within rtol 1e-6, residual within 1e-9 of the UNSHARDED oracle."""
    def tol(s):
        s.absTolerance = 1e-9
    s = M.LeastSquaresSettings(); tol(s)
    res, stats = sharded_solve(m_total, n, world, s)
    r0, x0 = res[0]
    for r, x in res[1:]:
        assert np.array_equal(x, x0) and (r.status, r.iterations, r.fCalls, r.residual, r.lambda_) == \
            (r0.status, r0.iterations, r0.fCalls, r0.residual, r0.lambda_)
    ro, xo = oracle_solve(oracle, m_total, n, tol)
    assert int(r0.status) >= 0 and ro.status >= 0
    assert np.allclose(x0, xo, rtol=1e-6, atol=1e-9), np.abs(x0 - xo).max()
    assert np.isclose(r0.residual, ro.residual, rtol=1e-9)
    # payloads of the three exchanges: packed n(n+1)/2 + n, sweep vector + trial sum 2n + 35, residual sums
    st = stats[0]
    assert st.allreduce_calls[0] >= 1 and st.allreduce_elems[0] == st.allreduce_calls[0] * (n * (n + 1) // 2 + n)
    if n <= 256 and st.jacobian_broyden:
        assert st.allreduce_elems[1] == st.allreduce_calls[1] * (2 * n + 35)
    assert st.allreduce_calls[1] + st.allreduce_calls[2] >= st.accepted + 1 and st.allreduce_elems[2] >= st.allreduce_calls[2]
    if n == 128:
        assert PAR.packed_length(n) == 8384 and st.allreduce_elems[0] % 8384 == 0
        assert st.allreduce_calls[1] == 0 or st.allreduce_elems[1] // st.allreduce_calls[1] == 291


def test_sharded_bounded_problem_runs_boxcqp_on_every_rank(oracle):
    """Bounds that the unconstrained steps violate: the BOXCQP active-set loop runs, replicated, on all eight shards."""
    m_total, n, world = 24000, 24, 8
    w = P.tanh_linear(m_total, n)
    lo = w["xstar"] - 0.5
    up = w["xstar"] + 0.5
    lo[::3] = w["xstar"][::3] + 0.02          # the minimiser is outside the box in every third coordinate
    x0 = np.clip(w["x0"], lo, up)

    def tol(s):
        s.absTolerance = 1e-9
    s = M.LeastSquaresSettings(); tol(s)
    res, stats = sharded_solve(m_total, n, world, s, lower=lo, upper=up, x0=x0)
    r0, xg = res[0]
    for r, x in res[1:]:
        assert np.array_equal(x, xg) and r.iterations == r0.iterations and r.residual == r0.residual
    assert stats[0].qp_active_set_passes > 0
    ro, xo = oracle_solve(oracle, m_total, n, tol, lower=lo, upper=up, x0=x0)
    assert int(r0.status) >= 0 and ro.status >= 0
    assert np.all(xg >= lo) and np.all(xg <= up)
    assert np.allclose(xg, xo, rtol=1e-6, atol=1e-9), np.abs(xg - xo).max()
    assert np.isclose(r0.residual, ro.residual, rtol=1e-9)


def test_one_rank_group_goes_through_every_collective():
    """A communicator with one rank still calls its all-reduce for each exchange (sum over one rank = identity):
    same bits as the solve without a communicator, and the payload lengths are the documented ones."""
    n = 128
    w = P.tanh_linear(30000, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    comms, close = PAR.local_group(1)
    st = M.Stats()
    r1, x1 = prob.solve(w["x0"], settings=s, comm=comms[0], stats=st, batched=True)
    close()
    r0, x0 = prob.solve(w["x0"], settings=s, batched=True)
    assert np.array_equal(x1, x0) and (r1.status, r1.iterations, r1.residual) == (r0.status, r0.iterations, r0.residual)
    assert st.allreduce_calls[0] == st.jacobian_full + st.jtj_resyncs and st.allreduce_elems[0] == st.allreduce_calls[0] * 8384
    assert st.allreduce_calls[1] == st.fused_rounds + st.jacobian_broyden - st.fused_passes and st.allreduce_elems[1] == 291 * st.allreduce_calls[1]
    assert st.allreduce_calls[2] >= 1


def test_two_host_threads_solve_different_problems_concurrently(oracle):
    """SURVEY 8b "Threading": concurrent solves on disjoint workspaces are legal. Two threads, different shapes (one takes
    the n <= 128 kernels, one the n = 256 ones, one is bounded), each on its own stream: identical to the solo runs."""
    w1, w2 = P.tanh_linear(60000, 128), P.tanh_linear(20000, 256)
    p1, p2 = W.TanhLinear(w1["A"], w1["b"]), W.TanhLinear(w2["A"], w2["b"])
    lo = w1["xstar"] - 0.3; up = w1["xstar"] + 0.3
    lo[::5] = w1["xstar"][::5] + 0.01
    x01 = np.clip(w1["x0"], lo, up)
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    solo1 = p1.solve(x01, l=lo, u=up, settings=s, batched=True)
    solo2 = p2.solve(w2["x0"], settings=s, batched=True)
    for _ in range(3):
        (ra, xa), (rb, xb) = run_threads([lambda: p1.solve(x01, l=lo, u=up, settings=s, batched=True),
                                          lambda: p2.solve(w2["x0"], settings=s, batched=True)])
        assert np.array_equal(xa, solo1[1]) and (ra.status, ra.iterations, ra.fCalls, ra.residual) == \
            (solo1[0].status, solo1[0].iterations, solo1[0].fCalls, solo1[0].residual)
        assert np.array_equal(xb, solo2[1]) and (rb.status, rb.iterations, rb.fCalls, rb.residual) == \
            (solo2[0].status, solo2[0].iterations, solo2[0].fCalls, solo2[0].residual)


def test_cfg4_full_size_eight_logical_shards_on_one_gpu():
    """BASELINE cfg 4 at FULL size on the one visible GPU: m = 8e6 x n = 256 fp64 (J alone is 16 GB; 288 GB of HBM hold the
    data set, eight shard workspaces and the unsharded one), rows split over EIGHT logical shards -- eight host threads,
    eight solver instances, the in-process communicator standing in for RCCL -- against the UNSHARDED solve of the same
    8e6-row problem on the same GPU. The oracle cannot run this size in test time (2.1 TFLOP per finite-difference refresh
    on the host); its parity at this shape is established 8-way at m = 40004 above. Here: every rank returns the same bits,
    and the sharded answer equals the unsharded one up to the order of the row reductions (x rtol 1e-9, residual 1e-11)."""
    m_shard, n, world = 1_000_000, 256, 8
    m_total = m_shard * world
    dA = M.DeviceBuffer(nbytes=m_total * n * 8, dtype=np.float64, shape=(m_total, n))
    db = M.DeviceBuffer(nbytes=m_total * 8, dtype=np.float64, shape=(m_total,))
    x0 = xstar = None
    for r in range(world):                                       # 2 GB of host memory at a time
        d = W.tanh_linear_data(m_shard, n, row_offset=r * m_shard)
        dA.upload_at(r * m_shard * n * 8, d["A"])
        db.upload_at(r * m_shard * 8, d["b"])
        x0, xstar = d["x0"], d["xstar"]
        del d
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-5
    comms, close = PAR.local_group(world)
    probs = [W.TanhLinearView(dA, db, r * m_shard, m_shard, n) for r in range(world)]
    stats = [M.Stats() for _ in range(world)]
    try:
        res = run_threads([(lambda r=r: probs[r].solve(x0, settings=s, comm=comms[r], stats=stats[r], batched=True)) for r in range(world)])
    finally:
        close()
    r0, xs = res[0]
    for r, x in res[1:]:
        assert np.array_equal(x, xs) and (r.status, r.iterations, r.fCalls, r.residual) == (r0.status, r0.iterations, r0.fCalls, r0.residual)
    st = stats[0]
    assert st.allreduce_calls[0] == st.jacobian_full + st.jtj_resyncs and st.allreduce_elems[0] == st.allreduce_calls[0] * PAR.packed_length(256)
    assert PAR.packed_length(256) == 33152 and st.allreduce_elems[1] == st.allreduce_calls[1] * (2 * 256 + 35)
    whole = W.TanhLinearView(dA, db, 0, m_total, n)
    ru, xu = whole.solve(x0, settings=s, batched=True)
    assert int(r0.status) >= 0 and r0.status == ru.status and r0.iterations == ru.iterations
    assert np.allclose(xs, xu, rtol=1e-9, atol=1e-12), np.abs(xs - xu).max()
    assert np.isclose(r0.residual, ru.residual, rtol=1e-11)
    assert np.abs(xu - xstar).max() < 1e-2                       # and it is the minimiser of the generating model
    dA.free(); db.free()


def test_record_and_replay_of_a_small_sharded_solve(oracle):
    """mir_lsq_comm_record / _recorded / _create_replay / _replay_rewind (a measurement tool): rank 0
    alone on the global trajectory returns the bits of the grouped solve, solve after solve (rewind), and an overflowing
    tape says so. A tape belongs to the round structure it was recorded with: the fused rounds exchange [sweep | trial sum]
    ONCE per trial, the one-by-one rounds (MIR_LSQ_VARIANT_NO_PIPELINE) the trial's sum and the sweep apart -- so each variant
    records and replays its own tape, and BOTH trajectories are the same bits (the exchanges of a fused round are
    unconditional: nothing speculative is ever missing from a tape)."""
    m_total, n, world = 40000, 32, 4
    w = P.tanh_linear(m_total, n)

    def shard(r):
        o, ml = PAR.row_shard(m_total, world, r)
        d = P.tanh_linear(ml, n, row_offset=o, m_total=m_total)
        return W.TanhLinear(d["A"], d["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    L = M.api.lib()
    prob = shard(0)
    outs, tapes = [], []
    for variant in (0, M.VARIANT_NO_PIPELINE):
        tape, rres, rx, _ = PAR.record_rank_tape(shard, world, w["x0"], settings=s, batched=True, variant=variant)
        assert rres.status >= 0 and tape.size > 0
        tapes.append(tape.size)
        comm = PAR.replay_comm(world, 0, tape)
        for rep in range(2):
            assert L.mir_lsq_comm_replay_rewind(C.c_void_p(comm)) == 0
            r1, x1 = prob.solve(w["x0"], settings=s, comm=comm, batched=True, variant=variant)
            outs.append((x1.tobytes(), int(r1.status), r1.iterations, r1.fCalls, r1.residual, r1.lambda_))
        assert outs[-1] == outs[-2] == (rx.tobytes(), int(rres.status), rres.iterations, rres.fCalls, rres.residual, rres.lambda_)
        L.mir_lsq_comm_destroy(C.c_void_p(comm))
    assert outs[0] == outs[2]                     # fused and one-by-one rounds: the same bits under sharding
    assert tapes[0] != tapes[1]
    # the global solve is the unsharded oracle's
    so = oracle.default_settings(); so.absTolerance = 1e-9
    ctx = oracle.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
    ro, xo = oracle.optimize(oracle.native_fn("wlc_tanh_linear_f"), m_total, w["x0"], settings=so, fctx=C.addressof(ctx))
    assert np.allclose(rx, xo, rtol=1e-6, atol=1e-9) and np.isclose(rres.residual, ro.residual, rtol=1e-9)
    # a tape that is too small reports (size_t)-1
    with pytest.raises(RuntimeError, match="overflow"):
        PAR.record_rank_tape(shard, world, w["x0"], capacity=16, settings=s, batched=True)


def test_replay_latency_model_delays_every_exchange_and_changes_no_bit():
    """mir_lsq_comm_replay_set_delay (the measurement tool behind `bench.py --replay-ranks R --replay-latency-us L`, DESIGN.md
    section 7): every replayed exchange first holds the stream L microseconds -- a MODEL of what an N-rank all-reduce costs.
    The solve's bits must not depend on it, its wall time must grow by at least L per exchange, and only a replay
    communicator accepts it."""
    import time
    m_total, n, world = 40000, 32, 4
    w = P.tanh_linear(m_total, n)

    def shard(r):
        o, ml = PAR.row_shard(m_total, world, r)
        d = P.tanh_linear(ml, n, row_offset=o, m_total=m_total)
        return W.TanhLinear(d["A"], d["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    L = M.api.lib()
    tape, rres, rx, _ = PAR.record_rank_tape(shard, world, w["x0"], settings=s, batched=True)
    prob = shard(0)
    comm = PAR.replay_comm(world, 0, tape)
    walls, outs, exchanges = [], [], 0
    for delay_us in (0, 2000):
        assert L.mir_lsq_comm_replay_set_delay(C.c_void_p(comm), delay_us) == 0
        best = None
        for rep in range(3):
            assert L.mir_lsq_comm_replay_rewind(C.c_void_p(comm)) == 0
            st = M.Stats()
            t0 = time.perf_counter()                               # (a solve returns when its result is on the host: synchronous)
            r1, x1 = prob.solve(w["x0"], settings=s, comm=comm, batched=True, stats=st)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            outs.append((x1.tobytes(), int(r1.status), r1.iterations, r1.fCalls, r1.residual, r1.lambda_))
            exchanges = sum(st.allreduce_calls)
        walls.append(best)
    assert len(set(outs)) == 1 and outs[0][0] == rx.tobytes() and exchanges >= 3
    assert walls[1] - walls[0] >= 0.9 * exchanges * 2000e-6, (walls, exchanges)      # every exchange was charged
    L.mir_lsq_comm_destroy(C.c_void_p(comm))
    comms, close = PAR.local_group(1)
    assert L.mir_lsq_comm_replay_set_delay(C.c_void_p(comms[0]), 10) == -1           # not a replay communicator
    close()
