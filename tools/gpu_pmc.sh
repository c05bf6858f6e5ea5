# PMC passes for the roofline "traffic" field (MI355X_MICROARCH.md, HBM section): one counter set per pass,
# FETCH_SIZE and WRITE_SIZE cannot share a pass; no trace domains besides --kernel-trace.
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf gpurun_out/pmc_$tag
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-side-legs > gpurun_out/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/pmc_$tag.log; }
  echo "pass $tag done"
done
