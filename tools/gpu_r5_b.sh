# round 5, second GPU call: conv tuning bits (variants build) A/B + PMC passes of the Dense_0 forward in the step and of the probe
mkdir -p gpurun_out/r5b && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5b
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
IDQN_HIP_LIB=$V IDQN_CONV_TUNE=7 timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py -x -q -m gpu > $O/fp_tune7.log 2>&1; echo "fp parity with TUNE=7 rc=$?"; tail -2 $O/fp_tune7.log
for rep in 1 2; do
bash tools/gpu_knobs.sh "IDQN_HIP_LIB=$V" "IDQN_HIP_LIB=$V IDQN_CONV_TUNE=1" "IDQN_HIP_LIB=$V IDQN_CONV_TUNE=2" "IDQN_HIP_LIB=$V IDQN_CONV_TUNE=4" "IDQN_HIP_LIB=$V IDQN_CONV_TUNE=3" "IDQN_HIP_LIB=$V IDQN_CONV_TUNE=7" "" 
done > $O/tune_ab.txt 2>&1; cat $O/tune_ab.txt
rocprofv3 -L > $O/counters.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/probes/ldsdma_stream_probe.hip -o /tmp/ldsdma_stream_probe > $O/probe_build.log 2>&1; echo "probe build rc=$?"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_BUSY_avr" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-30)
  rm -rf $O/pmc_step_$tag $O/pmc_probe_$tag
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_step_$tag -- python bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline > $O/pmc_step_$tag.log 2>&1 || { echo "step pass $tag failed"; tail -3 $O/pmc_step_$tag.log; }
  timeout -k 10 100 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_probe_$tag -- /tmp/ldsdma_stream_probe > $O/pmc_probe_$tag.log 2>&1 || { echo "probe pass $tag failed"; tail -3 $O/pmc_probe_$tag.log; }
  echo "pass $tag done"
done
python - <<'PY'
import csv, glob, collections, os
for kind in ("step", "probe"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in glob.glob("gpurun_out/r5b/pmc_%s_*/" % kind):
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        if "dense0_fwd3" in k or "k_reg" in k or "k_dma" in k:
            print(kind, k[:60], {n: round(sum(v) / len(v), 1) for n, v in sorted(c.items())}, "launches", max(len(v) for v in c.values()))
PY
