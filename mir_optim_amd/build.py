"""Build the HIP libraries in-tree with hipcc for gfx950 (cross-compiles without a GPU).

  lib/libmir_optim_amd.so            the solver + C ABI (include/mir_optim_amd.h)
  lib/libmir_optim_amd_workloads.so  device residual callbacks of the synthetic workloads

The solver is a dozen translation units (csrc/driver.h lists them): each is compiled to an object of its own under
build/obj/ -- in parallel, and only when it or one of the headers IT includes (the compiler's own dependency file, -MMD)
is newer -- and the objects are linked. A kernel edit costs the translation units that include that kernel's header
(usually one launch_*.hip), not the library.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build", "obj")
SOLVER_LIB = os.path.join(LIBDIR, "libmir_optim_amd.so")
WORKLOADS_LIB = os.path.join(LIBDIR, "libmir_optim_amd_workloads.so")

SOLVER_UNITS = ["abi.hip", "workspace.hip", "solver_loop.hip", "solver_jacobian.hip", "launch_jtj.hip", "launch_broyden.hip",
                "launch_solve_d.hip", "launch_solve_s.hip", "batched.hip", "comm.hip", "unit_entries.hip", "fit_spline.cpp"]
WORKLOAD_UNITS = ["workloads.hip", "workloads_gemm.hip", "workloads_resident.hip"]

_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("MIR_OPTIM_AMD_CXXFLAGS", "").split()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")
    return exe


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(s)) or os.path.getmtime(s) > t for s in sources)


def _deps(obj):
    """The files an object was compiled from, as the compiler recorded them (`-MMD -MF obj.d`): the translation unit and
    exactly the headers it includes -- an edit of a kernel header recompiles the units that include it and no others."""
    d = obj[:-2] + ".d"
    if not os.path.exists(d):
        return None
    toks = open(d).read().replace("\\\n", " ").split()
    return [t for t in toks[1:] if not t.endswith(":") and not t.startswith("/opt/") and not t.startswith("/usr/")]


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _compile_units(units, extra, force, verbose, jobs):
    objs, todo = [], []
    for u in units:
        src = os.path.join(CSRC, u)
        obj = os.path.join(OBJDIR, os.path.splitext(u)[0] + ".o")
        objs.append(obj)
        deps = _deps(obj)
        if force or deps is None or _stale(obj, [src] + deps):
            todo.append([_hipcc()] + _FLAGS + extra + ["-MMD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj])
    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(lambda c: _run(c, verbose), todo))
    return objs, bool(todo)


def build(force=False, verbose=False, jobs=None):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    # the flags are part of what an object depends on (a profiling build must not reuse the product objects)
    stamp = os.path.join(OBJDIR, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(_FLAGS):
        force = True
    objs, rebuilt = _compile_units(SOLVER_UNITS, [], force, verbose, jobs)
    open(stamp, "w").write(" ".join(_FLAGS))
    # the link takes the same flags as the compilations (-g, sanitizer or profiling flags from MIR_OPTIM_AMD_CXXFLAGS reach the .so)
    if force or rebuilt or _stale(SOLVER_LIB, objs):
        _run([_hipcc()] + _FLAGS + ["-shared", "-o", SOLVER_LIB] + objs + ["-ldl"], verbose)
    # the caller side: residual kernels of the synthetic workloads (the C entries + small kernels, the batched GEMM, and the
    # resident models, which are compiled against the device header of the resident-J solver like a caller's own model would be)
    wobjs, wrebuilt = _compile_units(WORKLOAD_UNITS, ["-fopenmp"], force, verbose, 3)     # OpenMP: host-side data generation / host residual
    if force or wrebuilt or _stale(WORKLOADS_LIB, wobjs):
        _run([_hipcc()] + _FLAGS + ["-shared", "-fopenmp", "-o", WORKLOADS_LIB] + wobjs, verbose)
    return SOLVER_LIB, WORKLOADS_LIB


if __name__ == "__main__":
    build(force=True, verbose=True)


def build_user_model_example(force=False, verbose=False):
    """tests/user_model/: a caller's own residual models compiled against include/mir_optim_amd_batched.hpp and
    include/mir_optim_amd_resident.hpp (the device headers of the batched fit and of the resident-J path) into a library of its own -- what a user of that header does. Built here so that it travels prebuilt."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "user_model", "user_model.hip")
    out = os.path.join(root, "tests", "user_model", "libuser_model.so")
    deps = [src, os.path.join(root, "include", "mir_optim_amd_batched.hpp"), os.path.join(root, "include", "mir_optim_amd.h"),
            os.path.join(CSRC, "batched_kernel.h"), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "solve_types.h"),
            os.path.join(root, "include", "mir_optim_amd_resident.hpp"), os.path.join(CSRC, "resident_kernel.h"),
            os.path.join(CSRC, "solve_wave16.h"), os.path.join(CSRC, "solve_kernel.h"), os.path.join(CSRC, "solve_lds.h")]
    if force or _stale(out, deps):
        _run([_hipcc()] + _FLAGS + ["-shared", "-I", os.path.join(root, "include"), "-o", out, src], verbose)
    return out
