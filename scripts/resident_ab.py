"""cfg 2 through the resident path with and without the look-ahead (MIR_LSQ_RESIDENT_NO_LOOKAHEAD): workgroup 0's stamps side by
side, best of a few runs. usage: python scripts/resident_ab.py [variant ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P
from mir_optim_amd import workloads as W

g = P.gauss_sum(100000, K=5)
r = W.Resident.gauss_sum(g["t"], g["data"], K=5)
variants = [int(v) for v in sys.argv[1:]] or [0, W.RESIDENT_NO_STAMPS, W.RESIDENT_NO_LOOKAHEAD, W.RESIDENT_NO_LOOKAHEAD | W.RESIDENT_NO_STAMPS]
for variant in variants:
    best = None
    wall = []
    for _ in range(9):
        t0 = time.perf_counter()
        res, x, st = r.solve(g["x0"], g["lower"], g["upper"], variant=variant)
        wall.append((time.perf_counter() - t0) * 1e6)
        if best is None or st["t_total"] < best["t_total"]:
            best = st
    st = best
    u = lambda k: st[k] / 100.0
    print(f"variant {variant}: host wall min {min(wall):.0f} us, kernel total {u('t_total'):.1f} us  rounds {st['rounds']} passes {st['passes']} look-ahead {st['lookahead_rejections']} elided {st['elided_evaluations']}  "
          f"per round {u('t_total') / st['rounds']:.2f}\n   worker {u('t_worker'):.1f} (eval {u('t_w_eval'):.1f} fd {u('t_w_fd'):.1f} products {u('t_w_prod'):.1f}) group {u('t_group'):.1f} "
          f"total-wait {u('t_total_wait'):.1f} solver {u('t_solver'):.1f} (solve body {u('t_solve_body'):.1f}, look {u('t_look'):.1f}, unpack {u('t_unpack'):.1f}, publish {u('t_publish'):.1f}) cmd-wait {u('t_cmd_wait'):.1f}  "
          f"status {res.status.name} it {res.iterations} fCalls {res.fCalls}", flush=True)
