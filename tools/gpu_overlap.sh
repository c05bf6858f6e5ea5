# overlap experiment (tools/probes/overlap_cumask.py): second stream unmasked / masked to n_b CUs, conv launches planned for 256 - n_b
export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_variants.so}  # the switches below exist in the variants build only
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
run() { IDQN_CUS=$1 timeout -k 10 170 python tools/probes/overlap_cumask.py $2 $3 200 2>&1 | grep -v Warning || exit 1; }
run 256 0 none && run 192 64 low && run 192 64 stride && run 160 96 low && run 128 128 low
