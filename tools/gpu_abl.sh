cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/abl
for v in hip abl1 abl2; do
  IDQN_HIP_LIB=i-dqn_amd/libidqn_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl/$v -o abl -- python3 bench.py --emulate-ranks 8 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/abl/$v.json 2> gpurun_out/abl/$v.err || exit 1
  f=$(find gpurun_out/abl/$v -name '*kernel_stats.csv' | head -1)
  echo "== $v: $(grep k_dense0_wgrad_alds $f | cut -d, -f1-4)"
done
