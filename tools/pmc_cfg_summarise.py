"""Box-side summary of tools/gpu_pmc_cfg.sh: per-kernel means of every counter of every pass + the kernel-stats table.

Run ON the GPU box (the raw rocprofv3 directories are too large to travel back): writes gpurun_out/pmcc_<tag>.json and
gpurun_out/stats_<tag>.csv, then the caller deletes the raw directories.  HBM bytes follow MI355X_MICROARCH.md (HBM
section): FETCH_SIZE / WRITE_SIZE are KiB, separate passes, FETCH_SIZE doubled on gfx950 for 16-B-per-lane streams.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(f"gpurun_out/pmcc_{tag}_*/"):
    files = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
st = glob.glob(f"gpurun_out/stats_{tag}/*/*_kernel_stats.csv")
if st:
    shutil.copy(st[0], f"gpurun_out/stats_{tag}.csv")
    for r in csv.DictReader(open(st[0])):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
out = {}
for k, c in agg.items():
    if not (k.startswith("k_") or k.startswith("void k_") or "::k_" in k):
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = {"launches": max(len(v) for v in c.values()), "counters_mean_per_launch": m}
    if k in dur:
        e["avg_us"] = dur[k][1]
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_read_bytes"] = 2.0 * m["FETCH_SIZE"] * 1024
        e["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
    g = m.get("GRBM_GUI_ACTIVE", 0)
    if g > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        e["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * g / 8)  # busy cycles summed over 1024 SIMDs, GUI_ACTIVE over 8 XCDs
    w = m.get("SQ_WAVE_CYCLES", 0)
    if w > 0:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if n in m:
                e[n.lower() + "_frac"] = m[n] / w
    if m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0) > 0:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    out[k] = e
json.dump(out, open(f"gpurun_out/pmcc_{tag}.json", "w"), indent=1, sort_keys=True)
print(f"pmcc_{tag}.json: {len(out)} kernels")
