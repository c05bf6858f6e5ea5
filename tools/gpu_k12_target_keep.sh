mkdir -p gpurun_out/r5pol && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5pol
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
for K in 1 2; do
for f in -1 $((2*K)) -1 $((2*K)); do
  cfg="IDQN_HIP_LIB=$V"; [ $f != -1 ] && cfg="$cfg IDQN_D0_FWD_NT_FROM=$f"
  env $cfg timeout -k 10 200 python bench.py --heads $K --steps 400 --warmup 30 --repeats 3 --no-cpu-baseline --no-side-legs > $O/ab.json 2> $O/ab.err || { echo "[$cfg] failed"; tail -5 $O/ab.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r5pol/ab.json')); print('K=$K nt_from=$f  %.4f ms' % d['ms_per_step'])"
done
done
