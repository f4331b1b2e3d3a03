"""Worker for the world_size-2 tests (launched by tests/test_distributed.py with RANK/WORLD_SIZE/MASTER_* set).
mode 'oracle': CPU only -- the oracle's sharded restatement over gloo (runs in the dev container).
mode 'gpu'   : the HIP path with the callback communicator over gloo; both ranks share the one visible GPU."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch.distributed as dist

import problems as P
from mir_optim_amd import parallel as PAR


def main():
    mode, m_total, n, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    off, m_local = PAR.row_shard(m_total, world, rank)
    w = P.tanh_linear(m_local, n, row_offset=off)
    ar = PAR.torch_allreduce_numpy(dist)
    if mode == "oracle":
        from oracle import oracle as O
        s = O.default_settings(); s.absTolerance = 1e-9
        ctx = O.TanhLinearCtx(w["A"].ctypes.data, w["b"].ctypes.data)
        res, x = O.optimize(O.native_fn("wlc_tanh_linear_f"), m_local, w["x0"], settings=s, fctx=C.addressof(ctx), allreduce=ar)
        status, iters, residual = O.STATUS[res.status], res.iterations, res.residual
    else:
        import mir_optim_amd as M
        from mir_optim_amd import workloads as W
        comm = PAR.HostAllreduceComm(world, rank, ar)
        prob = W.TanhLinear(w["A"], w["b"])
        s = M.LeastSquaresSettings(); s.absTolerance = 1e-9
        res, x = prob.solve(w["x0"], settings=s, comm=comm)      # the object: solve() re-raises what its callback recorded
        comm.close()
        status, iters, residual = res.status.name, res.iterations, res.residual
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(dict(status=status, iterations=iters, residual=float(residual), x=[float(v) for v in x]), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
