# de-phasing experiment: the fused Dense_0 update with its second / third residency slots delayed (IDQN_D0_STAGGER, ticks of 10 ns)
export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_variants.so}  # the switches below exist in the variants build only
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
bash tools/gpu_knobs.sh "" "IDQN_D0_STAGGER=300" "IDQN_D0_STAGGER=600" "IDQN_D0_STAGGER=1000" "" "IDQN_D0_STAGGER=1400" "IDQN_D0_STAGGER=600"
for st in 0 500 1000 1500 0 1000; do
  IDQN_D0_STAGGER=$st timeout -k 10 200 python bench.py --emulate-ranks 8 --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/emu_tmp.json 2> gpurun_out/emu_tmp.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/emu_tmp.json')); print('emulated N=8 stagger $st: %.1f us/step' % (d['ms_per_step']*1e3))"
done
