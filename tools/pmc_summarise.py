"""Summarises the rocprofv3 --pmc passes of tools/gpu_pmc.sh into profiles/ (run in the build container).

HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB,
collected in separate passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream, so it is doubled for the 16-B-per-lane streaming kernels; WRITE_SIZE is exact for 16-B stores.
"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
import os
for d in glob.glob("gpurun_out/pmc_*/"):
    files = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):  # newest pass only (older merges may linger in gpurun_out/)
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in agg.items():
    if not (k.startswith("k_") or k.startswith("void k_") or "::k_" in k):
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = {"launches": max(len(v) for v in c.values()), "counters_mean_per_launch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_read_bytes"] = 2.0 * m["FETCH_SIZE"] * 1024  # gfx950: doubled (see docstring)
        e["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        # MFMA busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
        e["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if "TCC_HIT_sum" in m and m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0) > 0:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    out[k] = e
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
dom = [k for k in out if "k_dense0_wgrad_pair" in k] or [k for k in out if "k_dense0_wgrad_rows" in k] or [k for k in out if "k_dense0_wgrad<true" in k]
if dom:
    import subprocess
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    # HBM bytes of ALL launches of one step: every step kernel's mean bytes x its launches per step (the dominant kernel runs
    # once per step; acting / sampling kernels of the bench's side legs are not step kernels)
    steps = out[dom[0]]["launches"]
    step_kernels = {k: e for k, e in out.items() if "hbm_bytes" in e and not any(x in k for x in ("k_act_", "k_sumtree", "k_replay", "k_per_", "k_sampler", "k_argmax"))}
    step_bytes = sum(e["hbm_bytes"] * e["launches"] / steps for e in step_kernels.values())
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    json.dump({"kernel": dom[0], "git": git, "csrc_sha256": bench.csrc_digest(), "hbm_bytes_per_launch": out[dom[0]]["hbm_bytes"],
               "step_hbm_bytes": step_bytes, "step_kernels": {k: round(e["hbm_bytes"] * e["launches"] / steps) for k, e in sorted(step_kernels.items())},
               "read": out[dom[0]]["hbm_read_bytes"], "write": out[dom[0]]["hbm_write_bytes"],
               "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                         "FETCH_SIZE doubled per MI355X_MICROARCH.md)"},
              open("profiles/pmc_traffic_latest.json", "w"), indent=1)
for k, e in sorted(out.items()):
    print(f"{k[:50]:50s}", {x: (round(y, 4) if isinstance(y, float) else y) for x, y in e.items() if x != "counters_mean_per_launch"})
