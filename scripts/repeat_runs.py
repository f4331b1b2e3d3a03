"""Run-after-run reproducibility of the entry points, in ONE fresh process (the first calls included: allocation paths, first-use
code loading): every repetition of a call must return the bits of the first. Found the stream-ordered-pool defect of the batched
launch in round 4 (about one launch in 150 returned wrong fits).  usage: repeat_runs.py [reps]"""
import hashlib, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mir_optim_amd as M
from mir_optim_amd import parallel as PAR, workloads as W
import problems as P
import test_gpu_batched as TB

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0


def digest(*arrs):
    h = hashlib.md5()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:12]


def check(name, fn, reps=REPS):
    global bad
    seen = {}
    for r in range(reps):
        seen.setdefault(fn(), []).append(r)
    ok = len(seen) == 1
    bad += not ok
    print(f"{'ok ' if ok else 'DIFFERS'}  {name}: {reps} runs, {len(seen)} distinct result(s)" + ("" if ok else f"  {[(k, v[:6]) for k, v in seen.items()]}"), flush=True)


# ---- batched fits: three models, shared and per-problem abscissae
for model, maker, count in ((M.MODEL_EXP_DECAY_PAD8, P.cfg5_pad8, 256), (M.MODEL_EXP_DECAY, TB.make_exp_decay, 192), (M.MODEL_EXP3_AFFINE, TB.make_exp3, 128)):
    t, data, truth, x0 = maker(count, 512)
    s = M.LeastSquaresSettings(np.float32)
    for per_problem in (False, True):
        tt = np.tile(t, (count, 1)) if per_problem else t
        def run(model=model, x0=x0, tt=tt, data=data, s=s):
            res, x = M.optimizeLeastSquaresBatched(model, x0, tt, data, settings=s)
            return digest(x, np.array([(int(r.status), r.iterations, r.fCalls) for r in res]))
        check(f"batched model {model} {'per-problem' if per_problem else 'shared'} abscissae", run)

# ---- standalone BOXCQP, both precisions, the three kernels (LDS, global factor, any n)
for n in (8, 100, 200, 300, 600):
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Pm = (Q * np.geomspace(1.0, 50.0, n)) @ Q.T
    q = rng.standard_normal(n) * 3
    xu = np.linalg.solve(Pm, -q)
    l = np.where(rng.random(n) < 0.3, xu + 0.05 * np.abs(xu) + 1e-2, -np.inf)
    u = np.where(rng.random(n) < 0.2, np.maximum(l, xu) + 0.5, np.inf)
    for dt in (np.float64, np.float32):
        def run(Pm=Pm, q=q, l=l, u=u, dt=dt):
            st, x, it = M.solveBoxQP(Pm, q, l, u, dtype=dt)
            return digest(x, np.array([int(st), it]))
        check(f"solveBoxQP n {n} {np.dtype(dt).name}", run, max(10, REPS // 2))

# ---- whole solves: device callbacks (bounded + unbounded, four solve kernels), the reference ABI with a python callback
for m, n, bounded in ((40000, 16, True), (30000, 64, False), (20000, 128, True), (9001, 192, True), (15000, 256, False), (4000, 300, True)):
    w = P.tanh_linear(m, n)
    prob = W.TanhLinear(w["A"], w["b"])
    lo = up = None
    x0 = w["x0"]
    if bounded:
        lo = w["xstar"] - 0.4; up = w["xstar"] + 0.4
        lo[::3] = w["xstar"][::3] + 0.02
        x0 = np.clip(x0, lo, up)
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    def run(prob=prob, x0=x0, lo=lo, up=up, s=s):
        r, x = prob.solve(x0, l=lo, u=up, settings=s, batched=True)
        return digest(x, np.array([int(r.status), r.iterations, r.fCalls, r.residual, r.lambda_]))
    check(f"device-callback solve {m} x {n} {'bounded' if bounded else 'unbounded'}", run, max(10, REPS // 2))

w = P.tanh_linear(300, 12)
def f(x, y): y[:] = np.tanh(w["A"] @ x) - w["b"]
def run():
    r, x = M.optimizeLeastSquares(f, 300, w["x0"].copy())
    return digest(x, np.array([int(r.status), r.iterations, r.fCalls, r.residual]))
check("reference ABI, python residual, 300 x 12", run, max(10, REPS // 2))

# ---- row shards in one process (2 and 4 host threads, one communicator each)
for world in (2, 4):
    m_total, n = 48000, 128
    s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
    probs = []
    for r in range(world):
        off, ml = PAR.row_shard(m_total, world, r)
        ww = P.tanh_linear(ml, n, row_offset=off)
        probs.append((W.TanhLinear(ww["A"], ww["b"]), ww))
    x0 = probs[0][1]["x0"]
    def run(world=world, probs=probs, x0=x0, s=s):
        comms, close = PAR.local_group(world)
        out = [None] * world
        def one(r):
            out[r] = probs[r][0].solve(x0, settings=s, comm=comms[r], batched=True)
        th = [threading.Thread(target=one, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join() for t in th]
        close()
        assert all(np.array_equal(out[r][1], out[0][1]) for r in range(world))
        return digest(out[0][1], np.array([int(out[0][0].status), out[0][0].iterations, out[0][0].fCalls, out[0][0].residual]))
    check(f"{world} in-process row shards, 48000 x 128", run, max(8, REPS // 4))

print("SUMMARY:", "all reproducible" if bad == 0 else f"{bad} entry point(s) NOT reproducible")
sys.exit(1 if bad else 0)
