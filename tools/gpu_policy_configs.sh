# round 5: "keep the online Dense_0 kernels of n heads on chip" (forward reads them and the update stores them with the default cache policy; IDQN_D0_KEEP=n,
# default n = as many heads as fit 84 MB) against every Dense_0 stream non-temporal (n = 0) over head counts and configurations; variants build, interleaved on one box.
# SETS="<bench args>|<n> <n> ..." entries separated by ';'
mkdir -p gpurun_out/r5pol && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5pol
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
IFS=';' read -ra SETS_A <<< "${SETS:---heads 6 --steps 400|0 5 6;--heads 7 --steps 300|0 5 7;--heads 8 --steps 300|0 4 5 8;--heads 12 --steps 200|0 5 12;--heads 64 --steps 60 --warmup 10|0 5;--heads 5 --steps 400|0 4 5}"
for set in "${SETS_A[@]}"; do
  args=${set%%|*}; ns=${set##*|}
  line=""
  for round in 1 2; do for n in $ns; do
    IDQN_HIP_LIB=$V IDQN_D0_KEEP=$n timeout -k 10 300 python bench.py $args --repeats 3 --no-cpu-baseline --no-side-legs > $O/c.json 2> $O/c.err || { echo "keep=$n [$args] failed"; tail -3 $O/c.err; continue; }
    line="$line $n:$(python -c "import json; print('%.4f' % json.load(open('gpurun_out/r5pol/c.json'))['ms_per_step'])")"
  done; done
  printf "%-44s keep n:ms  %s\n" "$args" "$line"
done
