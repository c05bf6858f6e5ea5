export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_variants.so}  # the switches below exist in the variants build only
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/twice
IDQN_D0_FWD_TWICE=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/twice -o tw -- python3 bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline > gpurun_out/twice/b.json 2> gpurun_out/twice/b.err || exit 1
python3 - <<'PY'
import csv, glob, statistics
f = sorted(glob.glob("gpurun_out/twice/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first, second = [], []
prev = None
for r in rows:
    if r["Kernel_Name"].startswith("k_dense0_fwd3"):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        (second if prev is not None and prev.startswith("k_dense0_fwd3") else first).append(d)
    prev = r["Kernel_Name"]
print("first launch  n=%d mean %.1f median %.1f" % (len(first), statistics.mean(first), statistics.median(first)))
print("second launch n=%d mean %.1f median %.1f" % (len(second), statistics.mean(second), statistics.median(second)))
PY
