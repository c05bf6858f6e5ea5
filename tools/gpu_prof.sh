# kernel-time profile of the bench step (rocprofv3 --kernel-trace --stats); prints per-kernel averages
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-cur}
rm -rf gpurun_out/prof_$tag; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-side-legs > gpurun_out/prof_$tag.log 2>&1
python - $tag <<'PY'
import csv,glob,sys,shutil
tag=sys.argv[1]
f=glob.glob(f'gpurun_out/prof_{tag}/*/*_kernel_stats.csv')
if f:
    shutil.copy(f[0], f'gpurun_out/{tag}_kernel_stats.csv')
    rows=list(csv.DictReader(open(f[0])))
    tot=sum(float(r['TotalDurationNs']) for r in rows if int(r['Calls'])>=100)
    for r in rows[:24]:
        print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
    print("sum of kernel time per step (us): %.1f"%(tot/120/1e3))
else:
    print(open(f'gpurun_out/prof_{tag}.log').read()[-3000:])
PY
