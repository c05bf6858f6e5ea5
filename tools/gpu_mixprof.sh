mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "IDQN_MIX_A=100 IDQN_MIX_B=0" "IDQN_MIX_A=0 IDQN_MIX_B=100" "IDQN_MIX_A=0 IDQN_MIX_B=0"; do
rm -rf gpurun_out/prof_mix; env IDQN_MIX=2 $cfg timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mix -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof_mix.log 2>&1
python - "$cfg" <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_mix/*/*_kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
print(sys.argv[1], {r['Name'][17:30]:round(float(r['AverageNs'])/1e3,1) for r in rows if 'k_mix_stage' in r['Name']})
PY
done
