# i-IQN step, Adam of Dense_0/kernel in the weight gradient's epilogue (k_iqn_d0_bwd_adam) against the two-split weight gradient + Adam
# pass (debug build, IDQN_IQN_ADAM_FUSE=0): tests, bench lines, rocprofv3 kernel stats, and the launch's timeline for the planned
# dispatch order and for every group's items back to back (IDQN_IQN_BWD_EARLY=99).  Outputs: gpurun_out/iiqn_ab/ (summary: ab.txt).
O=gpurun_out/iiqn_ab; mkdir -p $O; cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 300 python -m pytest tests/test_gpu_iqn.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log > $O/ab.txt
export IDQN_HIP_LIB=$GRAFT_REPO_ROOT/i-dqn_amd/libidqn_hip_debug.so
for rep in 1 2; do
  for f in 1 0; do
    IDQN_IQN_ADAM_FUSE=$f timeout -k 10 200 python bench.py --algo iiqn --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline > $O/bench_fuse${f}_$rep.json 2> $O/bench_fuse${f}_$rep.err || { tail -5 $O/bench_fuse${f}_$rep.err; exit 1; }
    python -c "import json; d=json.load(open('$O/bench_fuse${f}_$rep.json')); print('bench.py --algo iiqn, IDQN_IQN_ADAM_FUSE=$f, run $rep: %.1f steps/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" >> $O/ab.txt
  done
done
FUSE_LIST="1 0" bash tools/gpu_iiqn_prof.sh >> $O/ab.txt || exit 1
echo "--- timeline, planned order" >> $O/ab.txt
IDQN_PLAN_PRINT=1 timeout -k 10 200 python tools/probes/iqn_bwd_prof.py 2>&1 | grep -v "plan. fwd\|plan. wgrad\|amdgpu.ids" >> $O/ab.txt || exit 1
echo "--- timeline, every group's items back to back (IDQN_IQN_BWD_EARLY=99)" >> $O/ab.txt
IDQN_IQN_BWD_EARLY=99 timeout -k 10 200 python tools/probes/iqn_bwd_prof.py 2>&1 | grep -v "plan. \|amdgpu.ids" >> $O/ab.txt || exit 1
cat $O/ab.txt
