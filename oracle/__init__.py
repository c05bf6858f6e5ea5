"""CPU oracle for the i-DQN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from this package, and there only as the checker or as
the reported CPU baseline.  Nothing under ``i-dqn_amd/`` imports it; the product
path fails loudly when the HIP extension is missing instead of falling back.

What is restated, and what pins it
----------------------------------
* integer / fp64 path (``sumtree_ref``, ``samplers_ref``, ``replay_ref``):
  restatement of ``slimdqn/sample_collection/{sum_tree,samplers,replay_buffer}.py``.
  PINNED: checked against the reference's own known-answer tests
  (``tests/test_sum_tree.py``, ``tests/test_samplers.py``,
  ``tests/test_replay_buffer.py``) and against traces captured by importing the
  reference's ``sum_tree.py`` / ``samplers.py`` in the build container
  (``oracle/make_golden.py`` -> ``tests/golden/int_path_*.npz``).
* fp32 path (``qnet_ref``, ``torch_ref``): restatement of
  ``slimdqn/networks/idqn.py:13-24,96-131``, ``slimdqn/networks/dqn.py:60-92`` and
  ``slimdqn/networks/architectures/dqn.py:32-70`` plus the flax 0.10.2 / optax 0.2.4
  semantics they delegate to (``nn.Conv`` NHWC/HWIO ``padding="SAME"``, ``nn.Dense``
  ``x @ W + b``, ``optax.adam``).  jax / flax / optax are not installed in the build
  container (``ModuleNotFoundError`` -- an ordinary Python error, nothing was
  refused), so the reference's jitted step cannot be run here.
  PARITY UNPINNED by reference-captured vectors: the reference's own tests for
  this part (``tests/test_idqn.py:44-84``, ``tests/test_dqn.py:39-73``) hold no
  stored vectors, only the target / loss / argmax formulae, which are restated
  as known-answer tests.  The fp goldens under ``tests/golden/fp_path_*.json``
  come from the fp64 numpy restatement (``qnet_ref``) cross-checked against an
  independent torch-CPU autograd restatement (``torch_ref``).
"""
