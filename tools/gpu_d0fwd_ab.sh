# The Dense_0 forward inside the step: schedule variants and timing ablations of the variants build (rocprofv3 durations of the kernel;
# the ABL builds give WRONG results).  profiles/r5_d0fwd_ablations.txt is this script's output on two boxes.
mkdir -p gpurun_out/d0fwd && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/d0fwd
export IDQN_HIP_LIB=$PWD/i-dqn_amd/libidqn_hip_variants.so
IDQN_D0_FWD_XW=1 timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py -x -q -m gpu > $O/fp_xw.log 2>&1; echo "fp parity with XW rc=$?"; tail -1 $O/fp_xw.log
IDQN_D0_FWD_THREAD=1 timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py -x -q -m gpu > $O/fp_thread.log 2>&1; echo "fp parity with THREAD rc=$?"; tail -1 $O/fp_thread.log
for v in plain thread xw occ2 abl1 abl2 abl4 abl6 abl7 abl8 xw2 plain; do
  unset IDQN_D0_FWD_THREAD IDQN_D0_FWD_ABL IDQN_D0_FWD_XW IDQN_D0_OCC2 IDQN_D0_SPLITS
  case $v in plain) ;; thread) export IDQN_D0_FWD_THREAD=1;; xw) export IDQN_D0_FWD_XW=1;; xw2) export IDQN_D0_FWD_XW=2;;
             occ2) export IDQN_D0_OCC2=1 IDQN_D0_SPLITS=50;; abl*) export IDQN_D0_FWD_ABL=${v#abl};; esac
  rm -rf $O/prof_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline --no-side-legs > $O/prof_$v.log 2>&1 || { echo "$v failed"; tail -3 $O/prof_$v.log; continue; }
  f=$(find $O/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v: $(grep -E 'dense0_fwd3' $f | cut -d, -f1,4)   hidden $(grep -E 'k_hidden' $f | cut -d, -f4)"
  rm -rf $O/prof_$v
done
