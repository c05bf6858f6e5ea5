# round 5: "keep the online Dense_0 kernels on chip" (forward reads them and the update stores them with the default cache policy; IDQN_D0_KEEP=1)
# against every Dense_0 stream non-temporal (=0) over head counts and configurations; variants build, interleaved on one box
mkdir -p gpurun_out/r5pol && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5pol
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
for cfg in "--heads 1 --steps 400" "--heads 3 --steps 400" "--heads 5 --steps 400" "--heads 6 --steps 400" "--heads 7 --steps 300" "--heads 8 --steps 300" "--heads 12 --steps 200" "--heads 64 --steps 60 --warmup 10" "--batch 256 --steps 200 --warmup 20" "--batch 64 --steps 300" "--gpus 1 --force-dp --dp-streams inline --steps 300" "--emulate-ranks 8 --steps 200" "--actions 18 --steps 400"; do
  line=""
  for keep in 0 1 0 1; do
    IDQN_HIP_LIB=$V IDQN_D0_KEEP=$keep timeout -k 10 300 python bench.py $cfg --repeats 3 --no-cpu-baseline --no-side-legs > $O/c.json 2> $O/c.err || { echo "keep=$keep [$cfg] failed"; tail -3 $O/c.err; continue; }
    line="$line $(python -c "import json; print('%.4f' % json.load(open('gpurun_out/r5pol/c.json'))['ms_per_step'])")"
  done
  printf "%-56s keep 0 / 1 / 0 / 1: %s ms\n" "$cfg" "$line"
done
