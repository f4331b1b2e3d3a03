"""What the bench modules share: the peaks the roofline objects divide by (MI355X_MICROARCH.md), access to the committed
rocprofv3 PMC summaries (profiles/rNN/), and small helpers."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6        # dense f64 MFMA (= f64 vector) peak


def _round_no(path):
    import re
    m_ = re.search(r"profiles/r(\d+)/", path.replace(os.sep, "/"))
    return int(m_.group(1)) if m_ else -1


def pmc_file(m=1_000_000, n=128):
    """The newest (by round NUMBER) committed PMC summary for the per-GPU shape: profiles/rNN/pmc_traffic.json was taken
    at m = 1e6 x n = 128 (cfg 3), profiles/rNN/n256_pmc.json at m = 1e6 x n = 256 (cfg 4's per-GPU shape)."""
    import glob
    name = {(1_000_000, 128): "pmc_traffic.json", (1_000_000, 256): "n256_pmc.json"}.get((m, n))
    if name is None:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), key=_round_no)
    return files[-1] if files else None


def csrc_sha16(name):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "mir_optim_amd", "csrc", name), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_field(kernel, m, n, field):
    """HBM bytes per launch (or MFMA pipe utilisation) of `kernel` from the COMMITTED rocprofv3 PMC summary (made by
    scripts/pmc_summary.py / pmc_summary2.py from separate --pmc passes of this same command): PMC counters cannot be
    read from inside the timed run, so this is a stored measurement -- `traffic_source` in the JSON line says so. None
    if absent or if the per-GPU shape is not one of the profiled ones (m = 1e6 with n = 128 or 256)."""
    f = pmc_file(m, n)
    if f is None:
        return None
    try:
        ks = json.load(open(f))["kernels"]
        # template arguments added or dropped since (k_broyden_lr<double, 4, true> <-> <..., true, false>)
        if kernel not in ks:
            stem = kernel[:-1]
            kernel = next(k for k in ks if k.startswith(stem + ",") or stem.startswith(k[:-1] + ","))
        return ks[kernel][field]
    except (KeyError, ValueError, StopIteration):
        return None


def traffic_source(m, n):
    f = pmc_file(m, n)
    if f is None:
        return None
    return os.path.relpath(f,
                           ROOT) + (" (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command;"
                                    " not measured in this run)")


def flush_c_stdio():
    """RCCL prints its banner through C stdio, which is fully buffered on a pipe."""
    C.CDLL(None).fflush(None)
    sys.stdout.flush()


def describe_comm(api, comm):
    if not comm:
        return None
    buf = C.create_string_buffer(512)
    api.lib().mir_lsq_comm_describe(comm, buf, 512)
    return buf.value.decode()


def step_stats(ms):
    import statistics
    return [min(ms), statistics.median(ms), max(ms)] if ms else None
