# PMC passes + kernel stats of one bench configuration: tools/gpu_pmc_cfg.sh <tag> <bench args…>
# e.g. tools/gpu_pmc_cfg.sh b256 --batch 256   → gpurun_out/pmcc_b256_<pass>/, gpurun_out/stats_b256/
# One counter set per pass (FETCH_SIZE / WRITE_SIZE cannot share one), no trace domain besides --kernel-trace,
# the program directly behind `--`.  Summarise here with: python tools/pmc_summarise.py <tag> gpurun_out/pmcc_<tag>_
tag=$1; shift
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
B="bench.py --steps 12 --warmup 4 --repeats 1 --no-cpu-baseline --no-side-legs $*"
rm -rf gpurun_out/stats_$tag
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$tag -- python $B > gpurun_out/stats_$tag.log 2>&1 || { echo "stats failed"; tail -5 gpurun_out/stats_$tag.log; exit 1; }
echo "stats done"
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" \
            "TCC_HIT_sum TCC_MISS_sum"; do
  ptag=$(echo $pass | tr ' ' '_' | cut -c1-32)
  rm -rf gpurun_out/pmcc_${tag}_$ptag
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmcc_${tag}_$ptag -- python $B > gpurun_out/pmcc_${tag}_$ptag.log 2>&1 || { echo "pass $ptag failed"; tail -5 gpurun_out/pmcc_${tag}_$ptag.log; }
  echo "pass $ptag done"
done
python tools/pmc_cfg_summarise.py $tag && rm -rf gpurun_out/pmcc_${tag}_*/ gpurun_out/stats_$tag/
