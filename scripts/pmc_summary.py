"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes)
into profiles/<round>/pmc_traffic.json.  FETCH_SIZE is doubled (gfx950 reports half the bytes of a wide
coalesced streaming read; checked here against kernels with a known byte count: k_tanh_linear reads
8(mn + m) = 1.032 GB and reports 504 047 KB).  An optional third pass `pmc_MFMA` (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE)
adds the MFMA pipe utilisation per kernel: busy cycles (64 per v_mfma_f64_16x16x4_f64, summed over the 1024 SIMDs) over
GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3, hence / 8) x 1024 SIMDs.
usage: python scripts/pmc_summary.py gpurun_out profiles/r01"""
import collections, csv, glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
out = {}


def rows_of(path):
    """The rows of a counter table. k_broyden_lr has two kinds of launches since the fused rounds (round 6): the sweep, and -- when
    the record of the trial it runs behind rules a Broyden pass out -- a launch that only forms the trial's sum of squares
    (8 MB instead of 1 GB). The per-launch figures are the SWEEP's: its launches are the ones that last at least half as long
    as the kernel's longest (the table carries start / end timestamps per dispatch)."""
    rows = list(csv.DictReader(open(path)))
    longest = collections.defaultdict(float)
    for r in rows:
        if "k_broyden_lr" in r["Kernel_Name"]:
            longest[r["Kernel_Name"]] = max(longest[r["Kernel_Name"]], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return [r for r in rows if "k_broyden_lr" not in r["Kernel_Name"]
            or float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) >= 0.5 * longest[r["Kernel_Name"]]]

for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(src, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows_of(files[0]):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out.setdefault(k, {})[c + "_KB_per_launch"] = v / n
        out[k]["launches_" + c] = n
mf = glob.glob(os.path.join(src, "pmc_MFMA", "**", "*counter_collection.csv"), recursive=True)
if mf:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows_of(mf[0]):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["_rows"] += 1
    for k, d in agg.items():
        n = d["_rows"] / 4.0                                   # four counters per dispatch
        busy, gui = d["SQ_VALU_MFMA_BUSY_CYCLES"] / n, d["GRBM_GUI_ACTIVE"] / n
        e = out.setdefault(k, {})
        e["mfma_busy_cycles_per_launch"] = busy
        e["gui_active_cycles_per_xcd"] = gui / 8.0
        e["mfma_mops_f64_per_launch"] = d["SQ_INSTS_VALU_MFMA_MOPS_F64"] / n
        e["mfma_util"] = busy / (gui / 8.0 * 1024.0) if gui else 0.0
# optional SQ / LDS passes (pmc_SQ1, pmc_SQ2): per-launch averages of each counter, kept for the kernels that are not
# bandwidth-bound (k_lm_solve runs as ONE workgroup: wave cycles, VALU / LDS / SALU instructions, LDS bank conflicts)
for sq in ("pmc_SQ1", "pmc_SQ2"):
    files = glob.glob(os.path.join(src, sq, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in rows_of(files[0]):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    for k, d in agg.items():
        if "k_lm_solve" in k or "k_decide" in k or "k_lr_" in k or "k_broyden" in k or "k_jtj" in k:
            out.setdefault(k, {})["sq_per_launch"] = {**out.get(k, {}).get("sq_per_launch", {}), **{c: v / cnt[k][c] for c, v in d.items()}}
for k, d in out.items():
    f, w = d.get("FETCH_SIZE_KB_per_launch", 0.0), d.get("WRITE_SIZE_KB_per_launch", 0.0)
    d["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0          # FETCH_SIZE doubled: gfx950 correction
os.makedirs(dst, exist_ok=True)
json.dump({"note": "rocprofv3 --pmc, separate passes; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; "
                   "command: scripts/profile_bench.sh (bench.py --steps 4 --warmup 1 --no-cpu-baseline --survey-steps 0; cfg3, m=1e6, n=128)",
           "kernels": out}, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k]["hbm_bytes_per_launch"])[:8]:
    print(f"{k[:60]:60s} {out[k]['hbm_bytes_per_launch'] / 1e9:8.3f} GB/launch   MFMA util {out[k].get('mfma_util', 0.0):.3f}")
