# round 5, fourth GPU call: the Dense_0 forward with the W split threaded between the products (default now) against hipcc's own order
mkdir -p gpurun_out/r5d && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5d
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_configs.py -x -q -m gpu > $O/fp.log 2>&1; echo "fp + configs parity rc=$?"; tail -2 $O/fp.log
for rep in 1 2 3; do
bash tools/gpu_knobs.sh "IDQN_HIP_LIB=$V IDQN_D0_FWD_THREAD=0" "IDQN_HIP_LIB=$V" ""
done > $O/d0_thread_ab.txt 2>&1; cat $O/d0_thread_ab.txt
for v in plain threaded; do
  rm -rf $O/prof_$v
  if [ $v = plain ]; then export IDQN_HIP_LIB=$V IDQN_D0_FWD_THREAD=0; else unset IDQN_HIP_LIB IDQN_D0_FWD_PLAIN; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline --no-side-legs > $O/prof_$v.log 2>&1
  f=$(find $O/prof_$v -name '*kernel_stats.csv' | head -1); cp $f $O/${v}_kernel_stats.csv
  echo "== $v"; grep -E "dense0_fwd3|k_hidden|k_cfwd<3, 2, 3" $f | cut -d, -f1-4
done
rm -rf $O/prof_plain $O/prof_threaded
