mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export IDQN_CONV=bf16x3
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY"; do
  tag=c3_$(echo $pass | tr ' ' '_' | cut -c1-36)
  rm -rf gpurun_out/pmc_$tag
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/pmc_$tag.log; }
done
python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_c3_*/')):
    f=glob.glob(d+'*/*counter_collection.csv')
    if not f: print(d,"no counter csv"); continue
    rows=list(csv.DictReader(open(f[0])))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:30]][r['Counter_Name']].append(float(r['Counter_Value']))
    print("==",d)
    for k,c in agg.items():
        if 'conv' not in k: continue
        print(f"  {k:32s}", "  ".join(f"{n}={sum(v)/len(v):.4g}" for n,v in c.items()))
PY
