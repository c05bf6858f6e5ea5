# The factored update with the a3 tiles split in the kernel (default) against the round-4 plane copy of a3 (IDQN_DP_A3_PLANES=1, variants
# build): parity, then bench.py --emulate-ranks N interleaved on one box, and the B = 256 single-device line (same kernels)
mkdir -p gpurun_out/dpa3 && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/dpa3
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
timeout -k 10 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_configs.py tests/test_gpu_dp_two_ranks.py tests/test_gpu_dp_native.py -x -q -m gpu -k "factored or rccl or two_ranks or native or b256 or golden or config" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
run() { lbl="$1"; n="$2"; shift 2
  env "$@" timeout -k 10 200 python bench.py --emulate-ranks $n --steps 200 --warmup 50 --repeats 3 --no-cpu-baseline > $O/tmp.json 2> $O/tmp.err || { echo "[$lbl N=$n] failed"; tail -5 $O/tmp.err; return 1; }
  python - "$lbl" "$n" <<'PY'
import json, sys
d = json.load(open("gpurun_out/dpa3/tmp.json"))
print("%-34s N=%s  %.1f us/step  losses %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"] * 1e3, [round(x, 9) for x in d["final_losses"]]))
PY
}
for n in 8 4 2; do for rep in 1 2; do
  run "a3 split in the kernel (default)" $n IDQN_NONE=1
  run "a3 planes (IDQN_DP_A3_PLANES=1)" $n IDQN_HIP_LIB=$V IDQN_DP_A3_PLANES=1
done; done
for cfg in "IDQN_NONE=1" "IDQN_HIP_LIB=$V IDQN_DP_A3_PLANES=1"; do
  env $cfg timeout -k 10 200 python bench.py --batch 256 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline > $O/b256.json 2> $O/b256.err && python -c "
import json; d=json.load(open('gpurun_out/dpa3/b256.json')); k={x['launch']: x['us'] for x in d['kernels']}
print('B=256 [$cfg]: %.4f ms  factor planes %.1f  wgrad+adam %.1f' % (d['ms_per_step'], k.get('factor planes', 0), k.get('dense0 wgrad + adam', 0)))"
done
