"""Randomised differential sweep of the FUSED rounds against the one-by-one rounds (MIR_LSQ_VARIANT_NO_PIPELINE): the problems of
fuzz_parity.py (random shape, bounds incl. binding / pinned / tight boxes, starting point, settings), every callback flavour,
analytic Jacobian now and then, f64 and f32 -- both flows must return THE SAME BITS (x, status, counters, residual, lambda,
pass / rejection / QP statistics). Prints only the cases that differ.   python scripts/fuzz_fused.py [cases=600] [seed0=0] [shards]
`shards`: the same comparison on 2 ... 8 in-process row shards (one host thread each): a fused round exchanges [sweep | trial sum]
once where the one-by-one rounds exchange them apart -- every rank of both flows must hold the same bits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
import numpy as np

import mir_optim_amd as M
from mir_optim_amd import workloads as W
from fuzz_parity import case


def outcome(r, x, st):
    return (x.tobytes(), int(r.status), r.iterations, r.fCalls, r.gCalls, float(r.residual).hex(), float(r.lambda_).hex(), st.passes, st.accepted,
            st.rejected, st.step_guard_rejects, st.jacobian_full, st.jacobian_broyden, st.qp_active_set_passes, st.elided_evaluations)


def mid_case(seed):
    """fuzz_parity.case's recipe at 128 < n <= 256 (the solve with its factor in global memory: the fused round's rank-two term
    rides on that kernel's own copy of J^T J) -- the shapes case() draws stop at 129."""
    rng = np.random.default_rng(seed + 31337)
    n = int(rng.choice([129, 130, 144, 160, 191, 200, 255, 256]))
    m = int(n + rng.integers(0, 300)) if rng.random() < 0.6 else int(n + rng.integers(300, 1500))
    A = (2 * rng.random((m, n)) - 1) * np.sqrt(3.0 / n)
    xs = 2 * rng.random(n) - 1
    b = np.tanh(A @ xs) + 10.0 ** rng.integers(-6, -1) * (2 * rng.random(m) - 1)
    x0 = xs + 10.0 ** rng.integers(-3, 0) * (2 * rng.random(n) - 1)
    lo = np.full(n, -np.inf); up = np.full(n, np.inf)
    bounded = bool(rng.random() < 0.6)
    if bounded:
        sel = rng.random(n) < 0.5
        lo[sel] = xs[sel] - rng.random(sel.sum()) * 0.3 + (rng.random(sel.sum()) < 0.4) * 0.2
        sel = rng.random(n) < 0.5
        up[sel] = np.maximum(lo[sel] + 1e-3, xs[sel] + rng.random(sel.sum()) * 0.3 - (rng.random(sel.sum()) < 0.3) * 0.15)
    up = np.maximum(up, lo)
    x0 = np.clip(x0, lo, up)
    s = dict(maxIterations=int(rng.choice([1, 2, 5, 12, 40])), absTolerance=float(rng.choice([1e-3, 1e-6, 1e-9])),
             maxAge=int(rng.choice([0, 0, 1, 3])), gradTolerance=float(rng.choice([2.2e-16, 1e-8, 1e-3])))
    return dict(A=A, b=b, x0=x0, lo=lo, up=up, m=m, n=n, s=s, bounded=bounded)


def run_shards(c, sg, world, variant, lo, up):
    import threading
    from mir_optim_amd import parallel as PAR
    comms, close = PAR.local_group(world)
    probs = []
    for r in range(world):
        off, ml = PAR.row_shard(c["m"], world, r)
        probs.append(W.TanhLinear(c["A"][off:off + ml], c["b"][off:off + ml]))
    out, err, sts = [None] * world, [None] * world, [M.Stats() for _ in range(world)]

    def one(r):
        try:
            rr, xx = probs[r].solve(c["x0"], lo, up, settings=sg, comm=comms[r], stats=sts[r], batched=True, variant=variant)
            out[r] = outcome(rr, xx, sts[r])
        except BaseException as e:      # noqa: BLE001
            err[r] = e
    ts = [threading.Thread(target=one, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    hung = any(t.is_alive() for t in ts)
    close()
    for pb in probs:
        pb.dA.free(); pb.db.free()
    if hung or any(err):
        raise RuntimeError(f"sharded solve failed: hung={hung} {err}")
    return out, sts[0].fused_rounds


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    shards = len(sys.argv) > 3 and sys.argv[3] == "shards"
    tally = {"same": 0, "DIFFER": 0, "never fused": 0}
    for k in range(cases if shards else 0):
        c = mid_case(seed0 + k) if k % 4 == 3 else case(seed0 + k)
        rng = np.random.default_rng(seed0 + k + 4242)
        world = int(rng.choice([2, 3, 4, 8]))
        if c["m"] < 2 * world:
            continue
        sg = M.LeastSquaresSettings()
        for key, v in c["s"].items():
            setattr(sg, key, v)
        lo = c["lo"] if c["bounded"] else None
        up = c["up"] if c["bounded"] else None
        a, _ = run_shards(c, sg, world, M.VARIANT_NO_PIPELINE, lo, up)
        b, fused = run_shards(c, sg, world, 0, lo, up)
        if len(set(a)) == 1 and len(set(b)) == 1 and a[0] == b[0]:
            tally["same" if fused else "never fused"] += 1
        else:
            tally["DIFFER"] += 1
            print(f"DIFFER seed {seed0 + k} shards {world} m {c['m']} n {c['n']} bounded {c['bounded']} {c['s']}: ranks agree {len(set(a)) == 1} / {len(set(b)) == 1}\n"
                  f"   one-by-one {a[0][1:]}\n   fused      {b[0][1:]}", flush=True)
        if k % 25 == 0:
            print(f"... case {k}: {world} shards m {c['m']} n {c['n']} {tally}", flush=True)
    for k in range(0 if shards else cases):
        c = mid_case(seed0 + k) if (k % 4 == 3 or os.environ.get("FUZZ_MID") == "1") else case(seed0 + k)
        rng = np.random.default_rng(seed0 + k + 999)
        dtype = np.float32 if rng.random() < 0.25 else np.float64
        batched = [False, True, "rowmajor", "pointmajor"][int(rng.integers(0, 4))] if dtype == np.float64 else False
        analytic = bool(rng.random() < 0.2)
        extra = int(rng.choice([0, 0, 0, M.VARIANT_NO_SPECULATION, M.VARIANT_NO_NULL_SKIP, M.variant_lr_cap(int(rng.integers(1, 4)))]))
        sg = M.LeastSquaresSettings(dtype)
        for key, v in c["s"].items():
            setattr(sg, key, v)
        if dtype == np.float32:
            sg.absTolerance = max(sg.absTolerance, 1e-6); sg.gradTolerance = max(sg.gradTolerance, 1e-7)
        lo = c["lo"].astype(dtype) if c["bounded"] else None
        up = c["up"].astype(dtype) if c["bounded"] else None
        prob = W.TanhLinear(c["A"], c["b"], dtype=dtype)
        outs, fused = [], 0
        for variant in (M.VARIANT_NO_PIPELINE, 0):
            st = M.Stats()
            r, x = prob.solve(c["x0"].astype(dtype), lo, up, settings=sg, stats=st, batched=batched, analytic=analytic, variant=variant | extra)
            outs.append(outcome(r, x, st))
            fused = st.fused_rounds
        prob.dA.free(); prob.db.free()
        if outs[0] == outs[1]:
            tally["same" if fused else "never fused"] += 1
        else:
            tally["DIFFER"] += 1
            xa, xb = np.frombuffer(outs[0][0], dtype=dtype), np.frombuffer(outs[1][0], dtype=dtype)
            print(f"DIFFER seed {seed0 + k} {np.dtype(dtype).name} m {c['m']} n {c['n']} bounded {c['bounded']} batched {batched} g {analytic} extra {extra:#x} {c['s']}\n"
                  f"   one-by-one {outs[0][1:]}\n   fused      {outs[1][1:]}\n   max |dx| {np.abs(xa - xb).max():.3g}", flush=True)
        if k % 100 == 0:
            print(f"... case {k}: m {c['m']} n {c['n']} {np.dtype(dtype).name} {tally}", flush=True)
    print("summary:", tally)


if __name__ == "__main__":
    main()
