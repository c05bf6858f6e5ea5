# round 5, third GPU call: the Dense_0 forward at two waves per SIMD (variants build) + clean PMC passes of the forward in the step
# against the stand-alone reader of the same bytes in the same shape (250 x 256 threads, nt loads)
mkdir -p gpurun_out/r5c && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5c
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
IDQN_HIP_LIB=$V IDQN_D0_OCC2=1 IDQN_D0_SPLITS=50 timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py -x -q -m gpu > $O/fp_occ2.log 2>&1; echo "fp parity with OCC2 rc=$?"; tail -2 $O/fp_occ2.log
for rep in 1 2; do
bash tools/gpu_knobs.sh "IDQN_HIP_LIB=$V" "IDQN_HIP_LIB=$V IDQN_D0_OCC2=1 IDQN_D0_SPLITS=50" "IDQN_HIP_LIB=$V IDQN_D0_OCC2=1" "IDQN_HIP_LIB=$V IDQN_D0_OCC2=1 IDQN_D0_SPLITS=38" "IDQN_HIP_LIB=$V IDQN_D0_SPLITS=50" ""
done > $O/d0_occ2_ab.txt 2>&1; cat $O/d0_occ2_ab.txt
hipcc --offload-arch=gfx950 -O3 tools/probes/ldsdma_stream_probe.hip -o /tmp/ldsdma_stream_probe > $O/probe_build.log 2>&1; echo "probe build rc=$?"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_BUSY_avr" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_step_$tag -- python bench.py --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline --no-side-legs > $O/pmc_step_$tag.log 2>&1 || { echo "step pass $tag failed"; tail -3 $O/pmc_step_$tag.log; }
  timeout -k 10 100 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_probe_$tag -- /tmp/ldsdma_stream_probe > $O/pmc_probe_$tag.log 2>&1 || { echo "probe pass $tag failed"; tail -3 $O/pmc_probe_$tag.log; }
  echo "pass $tag done"
done
python - <<'PY' | tee gpurun_out/r5c/pmc_d0fwd_summary.txt
import csv, glob, collections
for kind in ("step", "probe"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r5c/pmc_%s_*/**/*counter_collection.csv" % kind, recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"][:44], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        if "dense0_fwd3" in k[0] or "k_reg" in k[0] or "k_dma" in k[0]:
            print(kind, k, {n: round(sum(v) / len(v), 1) for n, v in sorted(c.items())}, "n", max(len(v) for v in c.values()))
PY
rm -rf $O/pmc_step_* $O/pmc_probe_*
