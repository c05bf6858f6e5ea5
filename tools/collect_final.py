"""Copy the outputs of tools/gpu_final.sh (gpurun_out/final/) into profiles/<tag>_* under the names profiles/README.md indexes."""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "final"
src, dst = "gpurun_out/final", "profiles"
plain = {"bench.json": "bench.json", "bench_steps20.json": "bench_steps20.json", "bench_a18.json": "bench_a18.json", "bench_b256.json": "bench_b256.json",
         "bench_k64.json": "bench_k64.json", "bench_iiqn.json": "bench_iiqn.json", "kernel_stats.csv": "kernel_stats.csv",
         "iiqn_kernel_stats.csv": "iiqn_kernel_stats.csv", "loop.txt": "loop.txt", "bench_learner.json": "learner_bench.json", "bench_hp8.json": "hp8_bench.json"}
for a, b in plain.items():
    if os.path.exists(f"{src}/{a}"):
        shutil.copy(f"{src}/{a}", f"{dst}/{tag}_{b}")
for f in glob.glob(f"{src}/bench_dp1_*.json") + glob.glob(f"{src}/bench_emulate*.json"):
    shutil.copy(f, f"{dst}/{tag}_{os.path.basename(f)[len('bench_'):-len('.json')]}_bench.json")
rows = []
for f in sorted(glob.glob(f"{src}/bench_*=*.json")):
    x = json.load(open(f))
    rows.append("%-42s %7.1f steps/s  %.4f ms/step" % (os.path.basename(f)[len("bench_"):-len(".json")], x["value"], x["ms_per_step"]))
if rows:
    ref = {n: json.load(open(f"{src}/{n}"))["ms_per_step"] for n in ("bench.json", "bench_b256.json", "bench_k64.json") if os.path.exists(f"{src}/{n}")}
    with open(f"{dst}/{tag}_switches.txt", "w") as o:
        o.write("# shipped switches against the default on one box (tools/gpu_final.sh part c): steps/s, ms/step\n")
        o.write("# defaults of the same set (other calls, other boxes): " + "  ".join("%s %.4f" % (k[len("bench"):-len(".json")].strip("_") or "headline", v) for k, v in ref.items()) + " ms/step\n")
        o.write("\n".join(rows) + "\n")
print(open(f"{dst}/{tag}_switches.txt").read() if rows else "no switch runs")
