"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes)
into profiles/<round>/pmc_traffic.json.  FETCH_SIZE is doubled (gfx950 reports half the bytes of a wide
coalesced streaming read; checked here against kernels with a known byte count: k_tanh_linear reads
8(mn + m) = 1.032 GB and reports 504 047 KB).  usage: python scripts/pmc_summary.py gpurun_out profiles/r01"""
import collections, csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(src, f"pmc_{c}", "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out.setdefault(k, {})[c + "_KB_per_launch"] = v / n
        out[k]["launches_" + c] = n
for k, d in out.items():
    f, w = d.get("FETCH_SIZE_KB_per_launch", 0.0), d.get("WRITE_SIZE_KB_per_launch", 0.0)
    d["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0          # FETCH_SIZE doubled: gfx950 correction
os.makedirs(dst, exist_ok=True)
json.dump({"note": "rocprofv3 --pmc, separate passes; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; "
                   "command: bench.py --steps 2 --warmup 1 --no-cpu-baseline (cfg3, m=1e6, n=128)",
           "kernels": out}, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k]["hbm_bytes_per_launch"])[:8]:
    print(f"{k[:60]:60s} {out[k]['hbm_bytes_per_launch'] / 1e9:8.3f} GB/launch")
