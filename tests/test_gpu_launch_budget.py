"""Launches per round and the statistics that report them.

Single GPU, the product sequence (FUSED rounds: the tail behind a round's one trial residual is speculative sweep | reduce |
ONE kernel for decision + the pass's n x n side + the next solve, so the round that follows starts at its trial residual):
    refresh round   k_fd_points | fused FD J^T J | slab reduce (writes J^T J, J^T y) | solve | tail                     = 7
    Broyden round   (sweep, reduce, finish and solve ran in the previous round's tail) tail                             = 3
    re-solve round  solve | sums | decision (or solve | decision when every trial is a null step)                      <= 3
One-by-one rounds (MIR_LSQ_VARIANT_NO_PIPELINE; also what a trace selects):
    refresh round   k_fd_points | fused FD J^T J | slab reduce | solve | sums of squares (stage 1) | decision (stage 2 + walk) = 6
    Broyden round   sweep | reduce | finish | solve | sums | decision                                                   = 6
With a communicator the all-reduce sits between a reduction and its consumer: k_unpack_grad and k_lr_sumsq_final come back.
Reference loop: least_squares.d:972-1175 (the reductions are LS:1052, 1065, 1115)."""
import ctypes as C
import threading

import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import parallel as PAR
from mir_optim_amd import workloads as W
import problems as P

pytestmark = pytest.mark.gpu


def key(r, x):
    return (int(r.status), r.iterations, r.fCalls, r.gCalls, r.residual, r.lambda_, x.tobytes())


def counters(st):
    return (st.passes, st.accepted, st.rejected, st.step_guard_rejects, st.jacobian_full, st.jacobian_broyden,
            st.broyden_flushes, st.jtj_resyncs, st.elided_evaluations)


def test_bounded_gauss_sum_launch_counts():
    """cfg 2's family (bounded: BOXCQP kernel, n = 16): the launch budget of a round."""
    g = P.gauss_sum(100000, K=5)
    prob = W.Curve("gauss_sum", g["t"], g["data"])
    st = M.Stats()
    # (one-by-one rounds: the budget of the kernels taken apart)
    r, x = prob.solve(g["x0"], l=g["lower"], u=g["upper"], batched=True, stats=st, variant=M.VARIANT_NO_PIPELINE)
    assert r.status >= 0
    assert st.rounds[0] == st.jacobian_full and st.rounds[1] == st.jacobian_broyden
    # library launches per round, single GPU:
    #   Broyden round   sweep | reduce | finish | solve | sum of squares | decision                              = 6
    #   refresh round   FD points | fused FD J^T J | slab reduce (+ unpack) | solve | sums | decision            = 6
    #   re-solve round  solve | sums | decision (or solve | decision when every trial is a null step)           <= 3
    assert st.round_launches[1] == 6 * st.rounds[1]
    assert st.round_launches[2] <= 3 * st.rounds[2]
    assert st.round_launches[0] <= 8 * st.rounds[0]          # point-major panel + fill pass here; + k_reset_mu when LS:984 forces the refresh


def test_launch_budget_cfg3_shape():
    w = P.tanh_linear(50000, 128)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-5
    outs = []
    for variant in (M.VARIANT_NO_PIPELINE, 0):
        st = M.Stats()
        r, x = prob.solve(w["x0"], settings=s, batched=True, stats=st, variant=variant)
        assert r.status == M.LeastSquaresStatus.xConverged
        assert st.rounds[0] == st.jacobian_full >= 2 and st.rounds[1] == st.jacobian_broyden >= 2
        if variant:
            assert st.fused_rounds == 0
            assert st.round_launches[1] == 6 * st.rounds[1]          # sweep, reduce, finish, solve, sums, decision
            # FD points, k_jtj_fdp, slab reduce, solve, sums, decision (+ k_reset_mu in a refresh that LS:984-989 forces)
            assert 6 * st.rounds[0] <= st.round_launches[0] <= 6 * st.rounds[0] + 1
        else:
            # a round with one trial fuses its tail while first trials are being accepted; a Broyden pass that a fused tail
            # prepared costs its round 3 launches (speculative sweep, reduce, decision + finish + solve), one that follows a
            # refused prediction starts with its own sweep, reduce, finish, solve
            own = st.jacobian_broyden - st.fused_passes
            assert st.fused_rounds >= st.fused_passes >= st.jacobian_broyden - 1
            assert 3 * st.fused_passes + 6 * own <= st.round_launches[1] <= 3 * st.fused_passes + 7 * own
            assert 6 * st.rounds[0] <= st.round_launches[0] <= 7 * st.rounds[0] + 1
        assert st.library_launches == sum(st.round_launches) + 3  # + the sum of squares at entry (LS:955: two stages) and the state set-up
        outs.append((key(r, x), counters(st)))
    assert outs[0] == outs[1]


def test_one_rank_communicator_keeps_the_reductions_apart():
    """With a communicator the all-reduce sits between the reduction and its consumer: k_lr_finish / k_unpack_grad /
    k_decide_chain run behind the exchange. Same bits as without a communicator (one rank)."""
    n = 128
    w = P.tanh_linear(30000, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    r0, x0 = prob.solve(w["x0"], settings=s, batched=True)
    for variant in (M.VARIANT_NO_PIPELINE, 0):
        comms, close = PAR.local_group(1)
        st = M.Stats()
        r1, x1 = prob.solve(w["x0"], settings=s, comm=comms[0], stats=st, batched=True, variant=variant)
        close()
        assert key(r1, x1) == key(r0, x0)
        if variant:
            # Broyden round with a communicator: sweep | reduce | AR | finish | solve | sums x 2 | AR | decision = 7 launches
            assert st.round_launches[1] == 7 * st.rounds[1]
        else:
            # fused: the Broyden rounds a fused tail prepared cost 3 launches and ONE exchange (sweep | reduce | AR | decision +
            # finish + solve); one that follows a refused prediction starts with its own sweep, reduce, finish, solve (+ 4)
            own = st.jacobian_broyden - st.fused_passes
            assert st.fused_passes >= 1 and 3 * st.fused_passes + 6 * own <= st.round_launches[1] <= 3 * st.fused_passes + 8 * own


def test_eight_shards_agree_bitwise_run_after_run():
    world, m_total, n = 8, 64000, 128
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    outs = []
    for variant in (0, 0):
        comms, close = PAR.local_group(world)
        probs = []
        for r in range(world):
            off, ml = PAR.row_shard(m_total, world, r)
            w = P.tanh_linear(ml, n, row_offset=off)
            probs.append((W.TanhLinear(w["A"], w["b"]), w))
        x0 = probs[0][1]["x0"]
        res, err = [None] * world, [None] * world

        def one(r):
            try:
                res[r] = probs[r][0].solve(x0, settings=s, comm=comms[r], batched=True, variant=variant)
            except BaseException as e:   # noqa: BLE001
                err[r] = e
        ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(600)
        close()
        assert not any(t.is_alive() for t in ts) and not any(err), err
        outs.append([key(r, x) for r, x in res])
    assert len(set(outs[0])) == 1 and outs[0] == outs[1]


def test_repeated_solves_on_one_workspace_are_bit_identical(variant=0):
    """30 solves back to back on one workspace (device state, pinned mirrors and the event pool are reused), all bit-identical."""
    w = P.tanh_linear(40000, 64)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    ws = M.api.lib().mir_lsq_workspace_create(C.c_size_t(40000), C.c_size_t(64), C.c_size_t(8))
    assert ws
    try:
        ref = None
        for _ in range(30):
            r, x = prob.solve(w["x0"], settings=s, batched=True, workspace=ws, variant=variant)
            k = key(r, x)
            ref = ref or k
            assert k == ref
    finally:
        M.api.lib().mir_lsq_workspace_destroy(C.c_void_p(ws))


def test_stats_struct_is_versioned_by_size():
    """a caller compiled against an older header passes a smaller mir_lsq_stats. The library writes
    min(stats_size, sizeof) bytes; without a stats_size member the era of options.struct_size decides (header)."""
    w = P.tanh_linear(8000, 16)
    prob = W.TanhLinear(w["A"], w["b"])
    full = C.sizeof(M.Stats)
    for struct_size, stats_size, expect in [(64, None, 120), (72, None, 120), (80, None, 144), (88, None, 264),
                                            (96, 0, 264), (96, 152, 152), (96, full, full), (96, full + 64, full)]:
        buf = (C.c_ubyte * (full + 128))()
        C.memset(buf, 0, full + 128)
        for i in range(expect, full + 128):
            buf[i] = 0xA5                                            # canary behind what the caller's struct holds
        o = prob.options(batched=False)
        o.struct_size = struct_size
        o.stats = C.cast(buf, C.POINTER(M.Stats))
        o.stats_size = stats_size or 0
        r, x = M.api.optimizeLeastSquares(prob.f, prob.m, w["x0"].copy(), None, None, fContext=C.addressof(prob.ctx), options=o)
        assert r.status >= 0
        raw = bytes(buf)
        assert all(b == 0xA5 for b in raw[expect:]), f"struct_size {struct_size}: wrote past byte {expect}"
        st = M.Stats.from_buffer_copy(raw[:full])
        assert st.passes > 0 and st.accepted == r.iterations         # the part the caller has is filled in
