# round 5: timing ablations of the Dense_0 forward INSIDE the step (variants build, wrong results): rocprofv3 duration of the kernel
mkdir -p gpurun_out/r5e && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5e
export IDQN_HIP_LIB=$PWD/i-dqn_amd/libidqn_hip_variants.so
for v in plain thread abl1 abl2 abl4 abl6 abl7 abl8 plain; do
  unset IDQN_D0_FWD_PLAIN IDQN_D0_FWD_ABL
  case $v in plain) ;; thread) export IDQN_D0_FWD_THREAD=1;; abl*) export IDQN_D0_FWD_ABL=${v#abl};; esac
  rm -rf $O/prof_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline --no-side-legs > $O/prof_$v.log 2>&1 || { echo "$v failed"; tail -3 $O/prof_$v.log; continue; }
  f=$(find $O/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v: $(grep -E 'dense0_fwd3' $f | cut -d, -f1,4)   hidden $(grep -E 'k_hidden' $f | cut -d, -f4)"
  rm -rf $O/prof_$v
done
