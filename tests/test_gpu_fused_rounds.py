"""The fused rounds of the launch chain against the one-by-one rounds (MIR_LSQ_VARIANT_NO_PIPELINE), which are
the restatement of the reference's loop least_squares.d:972-1175 kernel by kernel: the next pass's Broyden sweep run
speculatively behind a trial residual (LS:1003-1006, 1052, 1065 ahead of the decision LS:1112-1161), the trial's sum of squares
(LS:1115) riding on it, and decision + n x n side + next solve in one kernel must give THE SAME BITS -- x, status, counters,
residual, lambda -- for every shape of the solve kernels (one wave n <= 16, LDS blocks n <= 128, global-memory factor n <= 256),
both element types, every callback flavour, with and without bounds, on trajectories that end in rejection tails and on ones
that stop at the iteration limit in the middle of a fused chain."""
import numpy as np
import pytest

import mir_optim_amd as M
from mir_optim_amd import workloads as W
import problems as P

pytestmark = pytest.mark.gpu


def outcome(r, x, st):
    return (x.tobytes(), int(r.status), r.iterations, r.fCalls, r.gCalls, r.residual, r.lambda_, st.passes, st.accepted, st.rejected,
            st.step_guard_rejects, st.jacobian_full, st.jacobian_broyden, st.broyden_flushes, st.qp_active_set_passes, st.elided_evaluations)


def both(prob, x0, lo=None, up=None, settings=None, extra_variant=0, **kw):
    """The same solve with the one-by-one rounds, then with the fused ones: ([outcome, outcome], [stats, stats])."""
    outs, stats = [], []
    for variant in (M.VARIANT_NO_PIPELINE, 0):
        st = M.Stats()
        r, x = prob.solve(x0, l=lo, u=up, settings=settings, stats=st, variant=variant | extra_variant, **kw)
        outs.append(outcome(r, x, st))
        stats.append(st)
    return outs, stats


@pytest.mark.parametrize("m,n,bounded,tol,batched", [
    (30000, 16, False, 1e-9, True), (30000, 16, True, 1e-9, True), (4000, 7, True, 1e-6, False), (20000, 32, False, 1e-9, True),
    (9000, 33, True, 1e-9, True), (30000, 64, True, 1e-6, "rowmajor"), (30000, 128, False, 1e-9, True), (20000, 128, True, 1e-5, False),
    (9999, 127, False, 1e-9, "pointmajor"), (9001, 192, True, 1e-9, True), (15000, 256, False, 1e-6, True), (5000, 200, False, 1e-12, True)])
def test_fused_rounds_give_the_bits_of_the_one_by_one_rounds(m, n, bounded, tol, batched):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = tol
    lo = up = None
    x0 = w["x0"]
    if bounded:
        lo = np.where(np.arange(n) % 3 == 0, w["xstar"] + 0.02, -np.inf)     # the minimiser violates a third of the bounds
        up = np.full(n, np.inf)
        x0 = np.maximum(x0, np.where(np.isfinite(lo), lo, -np.inf))
    outs, stats = both(prob, x0, lo, up, s, batched=batched)
    assert outs[0] == outs[1]
    assert stats[0].fused_rounds == 0 and stats[1].fused_rounds >= 1 and stats[1].fused_passes >= 1
    assert outs[0][1] >= 0


def test_fused_rounds_in_single_precision():
    """T = float (the reference is generic in T, LS:877): the workgroup solve's fused head, the float sweep and its sum of squares."""
    for m, n in ((20000, 24), (12000, 96)):
        w = P.tanh_linear(m, n)
        prob = W.TanhLinear(w["A"], w["b"], dtype=np.float32)
        s = M.LeastSquaresSettings(np.float32)
        outs, stats = both(prob, w["x0"].astype(np.float32), settings=s)
        assert outs[0] == outs[1] and stats[1].fused_passes >= 1 and outs[0][1] >= 0


def test_fused_chain_cut_by_the_iteration_limit_and_by_exits():
    """maxIterations reached in the middle of a chain of fused rounds (the decision's own test, LS:1175, must stop the kernel before
    the pass it would run ahead), fConverged through maxGoodResidual (LS:974), and the lambda > maxLambda exit (LS:979) at the end of
    a rejection tail: the same bits and counters as the one-by-one rounds."""
    w = P.tanh_linear(20000, 48)
    prob = W.TanhLinear(w["A"], w["b"])
    for it in (1, 2, 3, 5, 9):
        s = M.LeastSquaresSettings(); s.maxIterations = it
        outs, stats = both(prob, w["x0"], settings=s, batched=True)
        assert outs[0] == outs[1] and outs[0][2] <= it
    s = M.LeastSquaresSettings(); s.maxGoodResidual = 0.5                # reached after a few accepted steps
    outs, _ = both(prob, w["x0"], settings=s, batched=True)
    assert outs[0] == outs[1] and outs[0][1] == int(M.LeastSquaresStatus.fConverged)
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-15; s.maxLambda = 1e6   # the noisy end game runs into maxLambda
    outs, stats = both(prob, w["x0"], settings=s, batched=True)
    assert outs[0] == outs[1] and outs[0][9] >= 3                        # rejections were part of both trajectories


def test_analytic_jacobian_and_small_age_limits():
    """g given (maxAge 3, LS:945): refreshes every few passes, so fused tails alternate with rounds that cannot fuse (age); and
    maxAge = 1: ONE Broyden pass between two refreshes -- every other round's tail must decline to run a pass ahead."""
    w = P.tanh_linear(20000, 32)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    outs, stats = both(prob, w["x0"], settings=s, analytic=True)
    assert outs[0] == outs[1] and outs[0][4] >= 2 and stats[1].fused_passes >= 1
    s.maxAge = 1
    outs, stats = both(prob, w["x0"], settings=s, batched=True)
    assert outs[0] == outs[1] and stats[1].jacobian_full >= 3
    assert 1 <= stats[1].fused_passes <= stats[1].jacobian_broyden <= stats[1].jacobian_full


def test_pending_terms_cap_and_flush_inside_a_fused_chain():
    """The cap of pending rank-one terms (variant_lr_cap): when the next pass would have to flush them into J first, the round is
    not fused (the flush and the resynchronisation are kernels of their own), the rounds around it are."""
    w = P.tanh_linear(20000, 64)
    prob = W.TanhLinear(w["A"], w["b"])
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    flushed = 0
    for cap in (1, 2, 3):
        outs, stats = both(prob, w["x0"], settings=s, extra_variant=M.variant_lr_cap(cap), batched=True)
        assert outs[0] == outs[1]
        if stats[1].jacobian_broyden > cap:
            assert stats[1].broyden_flushes >= 1
            assert stats[1].fused_passes < stats[1].jacobian_broyden             # the passes behind a flush were not run ahead
            flushed += 1
    assert flushed >= 1


@pytest.mark.parametrize("m_total,n,world,bounded", [(60000, 64, 3, True), (64000, 128, 4, False), (30001, 128, 2, True), (40000, 16, 5, True)])
def test_fused_rounds_on_row_shards_give_the_bits_of_the_one_by_one_rounds(m_total, n, world, bounded):
    """Row shards over an in-process group (one host thread per shard): a fused round exchanges [sweep | trial sum] ONCE where the
    one-by-one rounds exchange the trial's sum and the sweep apart (LS:1115; LS:1052, 1065) -- every rank must end on the same
    bits in both, and on the same bits as every other rank."""
    import threading
    from mir_optim_amd import parallel as PAR
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    full = P.tanh_linear(m_total, n)
    lo = up = None
    x0 = full["x0"]
    if bounded:
        lo = np.where(np.arange(n) % 3 == 0, full["xstar"] + 0.02, -np.inf)
        up = np.full(n, np.inf)
        x0 = np.maximum(x0, np.where(np.isfinite(lo), lo, -np.inf))
    probs = []
    for r in range(world):
        off, ml = PAR.row_shard(m_total, world, r)
        w = P.tanh_linear(ml, n, row_offset=off, m_total=m_total)
        probs.append(W.TanhLinear(w["A"], w["b"]))
    per_flow = []
    for variant in (M.VARIANT_NO_PIPELINE, 0):
        comms, close = PAR.local_group(world)
        res, err, sts = [None] * world, [None] * world, [M.Stats() for _ in range(world)]

        def one(r):
            try:
                rr, xx = probs[r].solve(x0, l=lo, u=up, settings=s, comm=comms[r], stats=sts[r], batched=True, variant=variant)
                res[r] = outcome(rr, xx, sts[r])
            except BaseException as e:   # noqa: BLE001
                err[r] = e
        ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(600)
        close()
        assert not any(t.is_alive() for t in ts) and not any(err), err
        assert len(set(res)) == 1                                    # every rank: the same bits
        per_flow.append((res[0], sum(sts[0].allreduce_calls), sts[0].fused_rounds))
    assert per_flow[0][0] == per_flow[1][0]                           # and the same in both flows
    assert per_flow[1][2] >= 1 and per_flow[1][1] < per_flow[0][1]    # with fewer exchanges


@pytest.mark.parametrize("m,n", [(900, 129), (1500, 144), (700, 200), (4000, 64), (3000, 16)])
def test_gradient_test_exit_of_a_pass_run_ahead_keeps_its_rank_two_term(m, n):
    """A failed gradient test with an aged Jacobian (LS:1053-1062, quirk Q4) ends the pass before its solve -- but the Broyden update
    has happened and the NEXT pass solves with that J^T J again. The fused round's kernel runs the pass ahead; above n = 128 the
    rank-two term rides on the solve's own copy of J^T J, which the early exit never reaches: the term must go into memory
    there (it was lost: found by scripts/fuzz_fused.py, seeds 71, 307, 492, 575 at n = 129)."""
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    hit = 0
    for gtol in (1e-3, 1e-2, 3e-2):
        s = M.LeastSquaresSettings(); s.gradTolerance = gtol; s.absTolerance = 1e-9
        outs, stats = both(prob, w["x0"], settings=s, batched=False)
        assert outs[0] == outs[1], (gtol, outs[0][1:], outs[1][1:])
        hit += int(outs[0][1] == int(M.LeastSquaresStatus.gConverged) and stats[1].fused_rounds >= 1)
    assert hit >= 1


@pytest.mark.parametrize("seed,dtype", [(71, np.float32), (492, np.float32), (575, np.float64)])
def test_the_cases_the_differential_fuzzer_found(seed, dtype):
    """scripts/fuzz_fused.py, seeds 71 / 492 / 575 (n = 129, gradTolerance 1e-3): the fused rounds lost the rank-two term of a pass
    whose gradient test failed -- kept here as they were drawn."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from fuzz_parity import case
    c = case(seed)
    s = M.LeastSquaresSettings(dtype)
    for key, v in c["s"].items():
        setattr(s, key, v)
    if dtype == np.float32:
        s.absTolerance = max(s.absTolerance, 1e-6); s.gradTolerance = max(s.gradTolerance, 1e-7)
    lo = c["lo"].astype(dtype) if c["bounded"] else None
    up = c["up"].astype(dtype) if c["bounded"] else None
    prob = W.TanhLinear(c["A"], c["b"], dtype=dtype)
    outs, stats = both(prob, c["x0"].astype(dtype), lo, up, s, batched=False)
    assert c["n"] == 129 and outs[0] == outs[1] and stats[1].fused_rounds >= 1
