# timing ablations of the fused Dense_0 update (variant builds -DD0W_ABL=3 / 4 / 5: no phase 1 / no phase 3 / neither; wrong
# results): rocprofv3 duration of the kernel inside the default step
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/ablf
for v in hip abl3 abl4 abl5 hip; do
  IDQN_HIP_LIB=i-dqn_amd/libidqn_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ablf/$v -o abl -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/ablf/$v.json 2> gpurun_out/ablf/$v.err || exit 1
  f=$(find gpurun_out/ablf/$v -name '*kernel_stats.csv' | head -1)
  echo "== $v: $(grep 'k_dense0_wgrad<' $f | cut -d, -f1-4)  | finalize $(grep k_da3_finalize $f | cut -d, -f4)"
done
