import os, sys, time
ROOT = os.getcwd(); sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import numpy as np, torch
from slimdqn.sample_collection.replay_buffer import ReplayBuffer, TransitionElement
from slimdqn.sample_collection.samplers import UniformSamplingDistribution, PrioritizedSamplingDistribution
for name, sampler, kw in (("uniform", UniformSamplingDistribution(0), {}), ("prioritized", PrioritizedSamplingDistribution(0, 1_000_000, 0.6), {"priority": 1.0})):
    t0 = time.perf_counter()
    rb = ReplayBuffer(sampler, batch_size=32, max_capacity=1_000_000, stack_size=4, update_horizon=1, gamma=0.99)
    frames = np.random.default_rng(0).integers(0, 256, (64, 84, 84), dtype=np.uint8)
    n = 30000
    for i in range(n):
        rb.add(TransitionElement(frames[i % 64], i % 6, 1.0, i % 500 == 499, False), **kw)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(200): b = rb.sample()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: 1M-capacity buffer, {n} adds {(t1-t0)/n*1e6:.1f} us each (incl. allocation), sample {(t2-t1)/200*1e6:.1f} us, "
          f"ring {rb._frames.numel()/2**30:.2f} GiB, state {tuple(b.state.shape)}, mem {torch.cuda.memory_allocated()/2**30:.2f} GiB")
    del rb; torch.cuda.empty_cache()
