"""Build the HIP libraries in-tree with hipcc for gfx950 (cross-compiles without a GPU).

  lib/libmir_optim_amd.so            the solver + C ABI (include/mir_optim_amd.h)
  lib/libmir_optim_amd_workloads.so  device residual callbacks of the synthetic workloads
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SOLVER_LIB = os.path.join(LIBDIR, "libmir_optim_amd.so")
WORKLOADS_LIB = os.path.join(LIBDIR, "libmir_optim_amd_workloads.so")

_COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + os.environ.get("MIR_OPTIM_AMD_CXXFLAGS", "").split()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")
    return exe


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "mir_optim_amd.h"))
    jobs = [
        (SOLVER_LIB, [os.path.join(CSRC, "lm_driver.hip"), os.path.join(CSRC, "fit_spline.cpp")], hdrs, ["-ldl"]),
        (WORKLOADS_LIB, [os.path.join(CSRC, "workloads.hip")], [], ["-fopenmp"]),   # host-side data generation / host residual
    ]
    for target, srcs, deps, extra in jobs:
        if force or _stale(target, srcs + deps):
            cmd = [_hipcc()] + _COMMON + ["-o", target] + srcs + extra
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return SOLVER_LIB, WORKLOADS_LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
