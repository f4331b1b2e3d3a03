"""Summarise the passes of scripts/profile_any.sh: per kernel, per-launch averages of every counter that was collected, the
HBM bytes ((2 FETCH_SIZE + WRITE_SIZE) KB: the gfx950 correction of MI355X_MICROARCH.md) and the MFMA / VALU pipe utilisation
(busy cycles over GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  Copies the kernel-stats table next to it.
usage: python scripts/pmc_summary2.py gpurun_out/prof_<tag> profiles/r03 <prefix>"""
import collections, csv, glob, json, os, shutil, sys

src, dst, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)


def kname(s):
    return s.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


out = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(files[0])):
        k = kname(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    for k, cs in agg.items():
        for c, v in cs.items():
            out[k][c] = v / cnt[k][c]
            out[k]["launches_" + os.path.basename(d)[4:]] = cnt[k][c]
for k, d in out.items():
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
    gui = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                    # summed over the 8 XCDs by rocprofv3
    if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0)
    if gui and "SQ_ACTIVE_INST_VALU" in d:
        # SQ_ACTIVE_INST_VALU counts, per SIMD, the cycles a VALU instruction is executing (in units of 4 cycles: one wave64
        # instruction on a 16-lane SIMD); utilisation of the 1024 vector pipes over the kernel's active time
        d["valu_util"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / (gui * 1024.0)
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, prefix + "_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        k = kname(r["Name"])
        if k in out:
            out[k]["avg_ns"] = float(r["AverageNs"]); out[k]["calls"] = int(r["Calls"])
import hashlib
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mir_optim_amd", "csrc")
sha = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16] for f in sorted(os.listdir(csrc)) if f.endswith((".h", ".hip", ".inc"))}
json.dump({"note": "rocprofv3 --pmc, one counter set per pass (scripts/profile_any.sh); per-launch averages; hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KB",
           "source": src, "csrc_sha16": sha, "kernels": out}, open(os.path.join(dst, prefix + "_pmc.json"), "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k].get("avg_ns", 0) * out[k].get("calls", 0))[:10]:
    d = out[k]
    print(f"{k[:58]:58s} {d.get('avg_ns', 0) / 1e3:9.1f} us  hbm {d.get('hbm_bytes_per_launch', 0) / 1e9:7.3f} GB  mfma {d.get('mfma_util', 0):.3f}  valu {d.get('valu_util', 0):.3f}")
