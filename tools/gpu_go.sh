# build check + fp parity + conv phase probes + kernel profile, one GPU call:  bash tools/gpu_go.sh <tag> [roles...]
tag=${1:-cur}; shift
test -f i-dqn_amd/libidqn_hip.so || { echo "no library"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py -x -q 2>&1 | tail -5 || exit 1
for r in "$@"; do IDQN_CONV_PROF=$r timeout -k 10 100 python tools/probes/conv_prof.py 2>&1 | grep -v amdgpu.ids | tail -12; done
bash tools/gpu_prof.sh $tag
