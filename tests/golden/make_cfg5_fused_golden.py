"""Generates tests/golden/cfg5_fused_fits.json ON THE GPU BOX: the DEVICE's results (bit patterns) of the first 16 fits of BASELINE
cfg 5 (wave-per-problem kernel, padded exponential-decay model, m = 512, n = 8, fp32) together with the basis table the device
tabulated (sinf / cosf of the shared abscissae). The inputs are those of tests/problems.py:cfg5_pad8 (counter RNG: regenerated, not
stored). The CPU tier checks that oracle/lm_batched_fused.c reproduces these bits; the GPU tier that the kernel still does.

    python tests/golden/make_cfg5_fused_golden.py [out.json]      # needs a GPU
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import problems as P  # noqa: E402
import mir_optim_amd as M  # noqa: E402
from mir_optim_amd import api  # noqa: E402

FITS = 16


def device_fits(count=FITS, m=512, n=8):
    """(results as a structured array, x, basis table) of the first `count` cfg 5 problems on the device."""
    t, data, truth, x0 = P.cfg5_pad8(count)
    L = api.lib()
    s = M.LeastSquaresSettings(np.float32)
    dt_, dd, dx = api.DeviceBuffer(t), api.DeviceBuffer(data), api.DeviceBuffer(x0)
    dlo = api.DeviceBuffer(np.full(n, -np.inf, dtype=np.float32)); dup = api.DeviceBuffer(np.full(n, np.inf, dtype=np.float32))
    dres = api.DeviceBuffer(nbytes=count * 24, dtype=np.uint8, shape=(count * 24,))
    dbasis = api.DeviceBuffer(nbytes=m * 16, dtype=np.float32, shape=(m, 4))
    st = api.Stream()
    opt = api.BatchedOptions(stream=st.handle, basis=dbasis.ptr, basis_bytes=m * 16)
    rc = L.mir_lsq_batched_kernel_s(C.byref(s), count, m, M.MODEL_EXP_DECAY_PAD8, dx.ptr, dlo.ptr, dup.ptr, dt_.ptr, 0, dd.ptr, dres.ptr,
                                    C.byref(opt))
    assert rc == 0
    st.synchronize()
    raw = np.frombuffer(dres.download().tobytes(), dtype=np.dtype([("status", "<i4"), ("iterations", "<u4"), ("fCalls", "<u4"),
                                                                   ("gCalls", "<u4"), ("residual", "<u4"), ("lambda", "<u4")])).copy()
    x = dx.download().reshape(count, n).copy()
    basis = dbasis.download().reshape(m, 4).copy()
    for b in (dt_, dd, dx, dlo, dup, dres, dbasis):
        b.free()
    return raw, x, basis


def main():
    raw, x, basis = device_fits()
    out = {"_generator": "tests/golden/make_cfg5_fused_golden.py: DEVICE outputs (k_lm_batched<ModelExpDecayPad8>), bit patterns",
           "m": 512, "n": 8, "basis_bits": [int(v) for v in basis.view(np.uint32).ravel()], "fits": []}
    for k in range(FITS):
        out["fits"].append({"problem": k, "status": int(raw["status"][k]), "iterations": int(raw["iterations"][k]),
                            "fCalls": int(raw["fCalls"][k]), "residual_bits": int(raw["residual"][k]), "lambda_bits": int(raw["lambda"][k]),
                            "x_bits": [int(v) for v in x[k].view(np.uint32)]})
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "cfg5_fused_fits.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, [fit["iterations"] for fit in out["fits"]])


if __name__ == "__main__":
    main()
