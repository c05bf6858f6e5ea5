# rocprofv3 kernel stats of the i-IQN step, debug build: Adam in the weight gradient's epilogue (default) and IDQN_IQN_ADAM_FUSE=0.
O=gpurun_out/iiqn_prof; mkdir -p $O; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
export IDQN_HIP_LIB=$GRAFT_REPO_ROOT/i-dqn_amd/libidqn_hip_debug.so
for f in ${FUSE_LIST:-1 0}; do
  export IDQN_IQN_ADAM_FUSE=$f
  rm -rf $O/p$f; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$f -- python bench.py --algo iiqn --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $O/prof$f.log 2>&1 || { tail -5 $O/prof$f.log; exit 1; }
  cp $O/p$f/*/*_kernel_stats.csv $O/kernel_stats_fuse$f.csv; rm -rf $O/p$f
  python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_fuse$f.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows if int(r["Calls"])>=13)
print("fuse=$f")
for r in rows[:8]: print("  %-60s %4s %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
