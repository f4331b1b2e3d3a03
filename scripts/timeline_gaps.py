"""Where does the wall time of a solve go when the kernels are short? Reads a rocprofv3 --kernel-trace CSV of a bench run and
prints, for the steady part of the run: GPU busy time, idle time, and the idle time grouped by (kernel before -> kernel after).
usage: python scripts/timeline_gaps.py <dir or kernel_trace.csv> [skip_fraction]"""
import csv, glob, os, sys, collections

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * skip):]            # steady state: the timed loop at the end of the run


def short(name):
    name = name.split("(")[0]
    for p in ("void ", "mirlsq::", "(anonymous namespace)::"):
        name = name.replace(p, "")
    return name[:60]


busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    g = max(0, s1 - e0)
    key = (short(k0), short(k1))
    gaps[key][0] += g
    gaps[key][1] += 1
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows:
    per[short(k)][0] += e - s
    per[short(k)][1] += 1
nsolve = per.get("k_init_state<double>", [0, 1])[1] or 1
print(f"{path}\nkernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %)  idle {(span - busy) / 1e6:.2f} ms   solves ~{nsolve}")
print(f"per solve: span {span / nsolve / 1e3:.1f} us  busy {busy / nsolve / 1e3:.1f} us  idle {(span - busy) / nsolve / 1e3:.1f} us")
print("\nkernel time per solve (us), launches per solve, avg us")
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {t / nsolve / 1e3:9.1f}  {c / nsolve:6.1f}  {t / c / 1e3:8.1f}  {k}")
print("\nidle time per solve by kernel pair (us), count per solve, avg us")
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {t / nsolve / 1e3:9.1f}  {c / nsolve:6.1f}  {t / c / 1e3:8.1f}  {k[0]} -> {k[1]}")
