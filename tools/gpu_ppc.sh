mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
rm -rf gpurun_out/prof_ppc; env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ppc -- python bench.py --steps 120 --warmup 30 --no-cpu-baseline > gpurun_out/prof_ppc.log 2>&1
python - "$@" <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_ppc/*/*_kernel_stats.csv')
rows={r['Name']:float(r['AverageNs'])/1e3 for r in csv.DictReader(open(f[0]))}
g=lambda k: sum(v for n,v in rows.items() if k in n)
print(" ".join(sys.argv[1:]), "| wgrad<1,1> %.1f <1,2> %.1f <2,2> %.1f adam %.1f"%(g('k_conv_wgrad<1, 1>'),g('k_conv_wgrad<1, 2>'),g('k_conv_wgrad<2, 2>'),g('k_adam')))
PY
}
run X=0
run IDQN_PPC1=41
run IDQN_PPC1=25
run IDQN_PPC0=38
run IDQN_PPC0=41
