# round 5: the Dense_0 forward with wide activation loads through wave-private LDS (IDQN_D0_FWD_XW=1) against the plain kernel
mkdir -p gpurun_out/r5f && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5f
export IDQN_HIP_LIB=$PWD/i-dqn_amd/libidqn_hip_variants.so
IDQN_D0_FWD_XW=1 timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py -x -q -m gpu > $O/fp_xw.log 2>&1; echo "fp parity with XW rc=$?"; tail -2 $O/fp_xw.log
for v in plain xw xw2 abl2 plain xw; do
  unset IDQN_D0_FWD_PLAIN IDQN_D0_FWD_ABL IDQN_D0_FWD_XW
  case $v in plain) ;; xw) export IDQN_D0_FWD_XW=1;; xw2) export IDQN_D0_FWD_XW=2;; abl*) export IDQN_D0_FWD_ABL=${v#abl};; esac
  rm -rf $O/prof_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline --no-side-legs > $O/prof_$v.log 2>&1 || { echo "$v failed"; tail -3 $O/prof_$v.log; continue; }
  f=$(find $O/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v: $(grep -E 'dense0_fwd3' $f | cut -d, -f1,4)   hidden $(grep -E 'k_hidden' $f | cut -d, -f4)"
  rm -rf $O/prof_$v
done
