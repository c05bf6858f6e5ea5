import sys, os, json, numpy as np
sys.path[:0]=['/root/repo','/root/repo/i-dqn_amd','/root/repo/tests']
from test_gpu_fp_path import _agent
from slimdqn import _hip
for mode in ("bf16x3","f32"):
    os.environ["IDQN_CONV"]=mode
    agent, bs, rec, _ = _agent("cnn_atari_k64")
    K=agent._K
    agent._learn(bs[0], flags=_hip.F_GRADS_ONLY)
    G=agent._flat_grad()
    for leaf in ("Conv_0/kernel","Conv_1/kernel","Conv_2/kernel","Conv_0/bias"):
        d=rec["steps"][0]["leaves"][leaf]
        got=G[leaf].reshape(K,-1)[:,d["idx"]]; want=np.asarray(d["grad"])
        err=np.abs(got-want); i=np.unravel_index(err.argmax(), err.shape)
        print(mode, leaf, "max abs err %.3e at head %d probe %d: got %.6e want %.6e absmax %.3e" % (err.max(), i[0], i[1], got[i], want[i], np.asarray(d["grad_absmax"])[i[0]]))
