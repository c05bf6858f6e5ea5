/*
 * idqn_hip.h -- C ABI of the MI355X-native i-DQN hot path (libidqn_hip.so, gfx950 only).
 *
 * Plain pointers and sizes only; every pointer marked "dev" is a device (HBM) address, every
 * `stream` is a hipStream_t passed as void*.  All functions return 0 on success or a negative
 * IDQN_E_* code; nothing here falls back to a host implementation.
 *
 * The reference (theovincent/i-DQN) has no FFI: its seam is the Python object protocol used by
 * experiments/base/dqn.py.  Each entry point below names the reference operation it replaces
 * (file:line under the reference root); the Python mirror in i-dqn_amd/slimdqn binds them with
 * ctypes.
 */
#ifndef IDQN_HIP_H
#define IDQN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDQN_OK 0
#define IDQN_E_INVALID (-1)     /* bad argument / unsupported shape (message via idqn_last_error) */
#define IDQN_E_HIP (-2)         /* a HIP runtime call failed */
#define IDQN_E_RANGE (-3)       /* sumtree_query: a target is outside [0, root)   -> ValueError   */
#define IDQN_E_ASSERT (-4)      /* reference `assert` would have fired            -> AssertionError */

#define IDQN_ARCH_CNN 0         /* slimdqn/networks/architectures/dqn.py:39-53 */
#define IDQN_ARCH_FC 1          /* slimdqn/networks/architectures/dqn.py:61-63 */
#define IDQN_MAX_FEATURES 8
#define IDQN_MAX_LEAVES 24

const char* idqn_last_error(void);
int idqn_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Q-network / agent state.  Replaces iDQN.__init__ (slimdqn/networks/idqn.py:28-63) and
 * DQN.__init__ (slimdqn/networks/dqn.py:14-39) as far as device state goes.
 * ---------------------------------------------------------------------------------------- */
typedef struct idqn_config {
    int32_t arch;                         /* IDQN_ARCH_*                                        */
    int32_t n_heads;                      /* K  (n_networks, idqn.py:44; 1 for DQN)              */
    int32_t n_actions;                    /* A <= 32                                            */
    int32_t obs_h, obs_w, obs_c;          /* cnn: (84, 84, 4); fc: (dim, 1, 1)                    */
    int32_t n_features;
    int32_t features[IDQN_MAX_FEATURES];  /* cnn: [F0, F1, F2, F3]; fc: hidden widths            */
    int32_t max_batch;                    /* largest minibatch a call may pass                   */
    /* doubles: folded to f32 inside exactly like the reference's Python floats are by jit        */
    double learning_rate;                 /* optax.adam(lr, eps=adam_eps), idqn.py:52            */
    double adam_b1, adam_b2, adam_eps;
    double gamma_n;                       /* gamma ** update_horizon (a Python double), idqn.py:122 */
    int32_t n_quantiles;                  /* 0: i-DQN heads.  N > 0: i-IQN heads (extension, see idqn_iqn_learn_on_batch): N
                                             quantile fractions per sample for the online, the action-selection and the
                                             target pass; cnn only; adds the leaves Embed_0/kernel [64][F], Embed_0/bias [F] */
    int32_t reserved;
} idqn_config_t;

typedef struct idqn_leaf {
    char name[32];        /* flax leaf path, e.g. "Conv_0/kernel" (idqn.py:48-50 pytree)        */
    int64_t offset;       /* float offset of the leaf inside one head's slice of an arena        */
    int32_t ndim;
    int64_t shape[4];     /* per-head shape (HWIO for conv kernels, [in,out] for dense)          */
} idqn_leaf_t;

/* Arena layout: every parameter-like array (online, target, mu, nu) is [K][head_stride] floats; head k, leaf l
 * lives at arena + k*head_stride + leaf[l].offset.
 * The GRADIENT arena has K*head_stride + 64 floats arranged as two contiguous regions (two collectives in the
 * data-parallel step): [K][gP] all leaves except the cnn's Dense_0/kernel (same order; gP = head_stride - its padded
 * size), 64 floats reserved for the caller (the Python mirror keeps the K losses there), then [K][Dense_0/kernel].
 * For the fc architecture the second region is empty and gP = head_stride.                                       */
int idqn_layout(const idqn_config_t* cfg, int32_t* n_leaves, idqn_leaf_t* leaves /*[IDQN_MAX_LEAVES]*/,
                int64_t* head_stride);

typedef struct idqn_handle_s* idqn_handle_t;

/* The caller (PyTorch-ROCm tensors, storage only) owns the arenas; the handle owns activations
 * and scratch.  count: optax step counter per head (int32, idqn.py:53); losses: per-head loss of
 * the last step (idqn.py:109); cum_losses: f64 running sum == `cumulated_losses += losses`
 * (idqn.py:72) kept on the device so that no per-step host sync is needed.
 * Environment: IDQN_CONV = "bf16x3" (default) | "f32" selects, at creation, the conv arithmetic: f32-accurate
 * products on the bf16 matrix cores (exact three-way operand splits, csrc/convp.h) for every conv forward, data
 * gradient and weight gradient, or v_mfma_f32 everywhere.                                                          */
int idqn_create(const idqn_config_t* cfg, float* online_dev, float* target_dev, float* mu_dev, float* nu_dev,
                float* grad_dev, int32_t* count_dev, float* losses_dev, double* cum_losses_dev,
                idqn_handle_t* out);
int idqn_destroy(idqn_handle_t h);

/* flags for idqn_learn_on_batch */
#define IDQN_F_GRADS_ONLY 1u   /* stop after the gradients are in grad_dev (data-parallel: all-reduce, then idqn_apply_adam) */
#define IDQN_F_PROFILE 2u      /* bracket the dominant kernel with hipEvents (see idqn_profile_read) */
#define IDQN_F_PROFILE_ALL 16u /* one hipEvent after every launch of the step (see idqn_profile_table) */
#define IDQN_F_STOP_AFTER_DENSE0 4u  /* two-call backward, see idqn_backward_rest */
#define IDQN_F_STOP_BEFORE_DENSE0_WGRAD 8u  /* factored data-parallel step, see idqn_finish_step_factored */

/* i-IQN heads -- BASELINE config 3, a LABELLED EXTENSION: the reference snapshot has no quantile code (its README.md:3,10
 * names i-IQN and points at another repository), so nothing here replaces a reference function; oracle/iqn_ref.py states
 * the algorithm (implicit quantile network, Dabney et al. 2018, on the reference's conv trunk; the reference's chain of K
 * heads, idqn.py:13-24,96-109) and parity is pinned to that restatement only.
 * One gradient step of K heads on a minibatch of batch <= 32 samples: per head N online fractions, N action-selection
 * fractions and N target fractions per sample, tau_dev = float32 [K][3][N][batch] in (0, 1) (the host draws them);
 * quantile Huber loss (kappa = 1), sum over the online and mean over the target fractions, mean over the batch; Adam on
 * every leaf; count += 1, losses written, cum_losses accumulated -- as idqn_learn_on_batch.  flags: IDQN_F_PROFILE / IDQN_F_PROFILE_ALL only.
 * Refused (IDQN_ERR_*) while idqn_set_per_buffers has buffers set: the quantile loss takes no importance weights. */
int idqn_iqn_learn_on_batch(idqn_handle_t h, const void* state_dev, const void* next_state_dev,
                            const int32_t* action_dev, const float* reward_dev, const uint8_t* terminal_dev,
                            const float* tau_dev, int32_t batch, uint32_t flags, void* stream);
/* Acting rule of IQN for n <= 32 states: q[a] = mean over the N fractions tau_dev [N][n] of Z(s, tau)[a] of head `head`
 * (which = 0 online / 1 target) -> q_out_dev [n][A]; action_out_dev [n] (may be NULL) = argmax, first maximum on ties. */
int idqn_iqn_q_values(idqn_handle_t h, int32_t which, int32_t head, const void* states_dev, int32_t n,
                      const float* tau_dev, float* q_out_dev, int32_t* action_out_dev, void* stream);

/* iDQN.learn_on_batch (idqn.py:96-109) == DQN.learn_on_batch (dqn.py:60-73) for K == 1:
 * 2K forwards, TD target (idqn.py:120-124), squared loss mean over the batch (idqn.py:111-118),
 * K backwards, Adam, count += 1, losses written, cum_losses accumulated.
 * state / next_state: cnn: uint8 [B][H][W][C] (NHWC, the reference's stacked batch, replay_buffer.py:229);
 *                     fc : float32 [B][dim].
 * batch_mean_divisor: the B of the `.mean()` -- pass the GLOBAL batch size when the minibatch is
 * sharded over ranks so that summed shard gradients equal the full-batch gradient.                */
int idqn_learn_on_batch(idqn_handle_t h, const void* state_dev, const void* next_state_dev,
                        const int32_t* action_dev, const float* reward_dev, const uint8_t* terminal_dev,
                        int32_t batch, int32_t batch_mean_divisor, uint32_t flags, void* stream);
/* iDQN.update_online_params (idqn.py:65-72) = ReplayBuffer.sample() (replay_buffer.py:215-230) + learn_on_batch, as ONE call on
 * the HBM frame ring: the stacked gather of replay_gather_stacked (frames of `stack` consecutive transitions, zero frames before
 * an episode start, replay_buffer.py:119-137,223-229) happens inside the step's staging launch, so the sampled minibatch is
 * never materialised and the slots travel as kernel arguments -- no upload, no gather launch.  Same arithmetic and the same
 * results, bit for bit, as replay_gather_stacked followed by idqn_learn_on_batch on its outputs.
 *   frame_ring_dev  uint8 [n_frames][frame_bytes]          (replay_add_frame)
 *   rows_dev        int32 [capacity][8] element rows       (replay_gather_stacked documents the row)
 *   slots_host      int32 [batch] sampled element slots (= key % capacity), HOST memory, read before the call returns
 * Restrictions (IDQN_ERR_INVALID otherwise; callers then gather and call idqn_learn_on_batch): cnn arch on the plane conv
 * path, uint8 frames with frame_bytes == obs_h * obs_w and a multiple of 16, stack == obs_c == 4, batch <= 256.            */
int idqn_learn_on_replay(idqn_handle_t h, const uint8_t* frame_ring_dev, int64_t n_frames, int64_t frame_bytes,
                         const int32_t* rows_dev, const int32_t* slots_host, int32_t batch, int32_t stack,
                         int32_t batch_mean_divisor, uint32_t flags, void* stream);
/* Data-parallel overlap: with IDQN_F_STOP_AFTER_DENSE0 (implies gradients only) idqn_learn_on_batch returns
 * once the Dense_0 weight gradient -- 98 % of the gradient bytes, produced first in the backward pass -- is queued;
 * the caller starts all-reducing that slice (RCCL, async) and calls idqn_backward_rest for the conv backward,
 * which then runs concurrently with the collective.                                                               */
int idqn_backward_rest(idqn_handle_t h, void* stream);
/* Factored data-parallel step.  The Dense_0/kernel gradient (98 % of all gradient bytes) is the outer product
 * a3^T . dh over the samples -- rank <= global batch -- so ranks exchange the FACTORS (K x F x 32 + K x J x 32 floats
 * per 32-sample block, ~5.3 MB at K=5) instead of all-reducing the 79 MB product, and every rank runs the fused
 * weight-gradient + Adam kernel over the global batch:
 *   idqn_learn_on_batch(.., IDQN_F_STOP_BEFORE_DENSE0_WGRAD)   forward, head, Dense_0 data gradient
 *   idqn_export_dense0_factors   async copies of this rank's a3 [K][nb][F*32] and dh [K][nb][J*32]
 *   (all-gather both; meanwhile idqn_backward_rest = conv backward; all-reduce the small-leaf gradient region)
 *   idqn_finish_step_factored    phase IDQN_FACTORED_DENSE0: fused Dense_0 update from the gathered factors (needs
 *                                only the gather, so it can run while the small-leaf all-reduce is still in flight);
 *                                phase IDQN_FACTORED_REST: Adam on every other leaf from grad_dev, count += 1,
 *                                cum_losses += losses.  `phases` = either or both (3), Dense_0 first.
 * Block bb of the gathered buffers lives at (bb / nb_inner) * outer + head * head_stride + (bb % nb_inner) * inner
 * (strides in floats); a plain all_gather of the exported buffers gives outer = K*nb*X, head = nb*X, inner = X.   */
int idqn_export_dense0_factors(idqn_handle_t h, float* a3_out_dev, float* dh_out_dev, void* stream);
/* The same two factors WITHOUT a copy: the library keeps dL/dh [K][nb][J*32] directly in front of the online nets' a3
 * [K][nb][F*32], so one contiguous run of n_dh + n_a3 floats at *factors_dev is what a rank contributes to the
 * all-gather (valid from the IDQN_F_STOP_BEFORE_DENSE0_WGRAD call until the next step; ordered behind that call's
 * stream).  In the gathered buffer dh_all = gathered, a3_all = gathered + n_dh, outer stride = n_dh + n_a3.          */
int idqn_dense0_factors(idqn_handle_t h, float** factors_dev, int64_t* n_dh, int64_t* n_a3);
#define IDQN_FACTORED_DENSE0 1u
#define IDQN_FACTORED_REST 2u
int idqn_finish_step_factored(idqn_handle_t h, const float* a3_all_dev, const float* dh_all_dev, int32_t nb_total,
                              int32_t nb_inner, int64_t a3_outer, int64_t a3_head, int64_t a3_inner,
                              int64_t dh_outer, int64_t dh_head, int64_t dh_inner, uint32_t phases, void* stream);
/* Second half of the data-parallel step: Adam from grad_dev, count += 1. */
int idqn_apply_adam(idqn_handle_t h, void* stream);

/* ------------------------------------------------------------------------------------------
 * The data-parallel step with its collectives inside the library (RCCL over xGMI, one process per GPU).  No reference
 * counterpart: iDQN.learn_on_batch (idqn.py:96-109) is single-device; this shards its minibatch mean (idqn.py:111-112).
 * RCCL is resolved at run time (librccl.so.1), the single-GPU entry points do not depend on it.
 * ---------------------------------------------------------------------------------------- */
typedef struct idqn_dp_s* idqn_dp_t;
#define IDQN_DP_UNIQUE_ID_BYTES 128
#define IDQN_DP_SIDE_STREAM 1u /* collectives on a stream of the library's own, ordered by events (the all-gather then runs under
                                  the conv backward); 0: on the caller's stream, in program order, no cross-stream hand-over */
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to every rank (any channel: a file, MPI, torch.distributed). */
int idqn_dp_unique_id(void* id_out /*[IDQN_DP_UNIQUE_ID_BYTES]*/);
/* ncclCommInitRank on the calling thread's current device: collective over the `world` ranks of the job. */
int idqn_dp_create(idqn_handle_t h, const void* unique_id, int32_t rank, int32_t world, uint32_t flags, idqn_dp_t* out);
/* The same around a communicator the caller already owns (an ncclComm_t passed as void*; not destroyed by idqn_dp_destroy). */
int idqn_dp_create_from_comm(idqn_handle_t h, void* nccl_comm, uint32_t flags, idqn_dp_t* out);
/* Destroys the step object (and the communicator idqn_dp_create made).  Call it BEFORE idqn_destroy of the handle it was built on. */
int idqn_dp_destroy(idqn_dp_t dp);
/* rank, world, bytes a rank contributes to the all-gather per 32-sample block of its shard, bytes of the all-reduce (any NULL). */
int idqn_dp_info(idqn_dp_t dp, int32_t* rank, int32_t* world, int64_t* gather_bytes_per_rank, int64_t* allreduce_bytes);
/* One global gradient step: this rank's `batch`-sample shard of a global batch of global_batch = world * batch samples.
 * Enqueues idqn_learn_on_batch(IDQN_F_STOP_BEFORE_DENSE0_WGRAD) -> ncclAllGather of the [dL/dh | a3] run -> conv backward
 * (idqn_backward_rest) -> ncclAllReduce(sum) of the small-leaf gradient region and the K losses -> the fused Dense_0 update
 * over the gathered factors -> Adam on every other leaf, count += 1, cum_losses += losses (idqn_finish_step_factored).
 * Nothing synchronises with the host.  losses_dev then holds the loss of the GLOBAL batch.  flags: IDQN_F_PROFILE(_ALL). */
int idqn_dp_step(idqn_dp_t dp, const void* state_dev, const void* next_state_dev, const int32_t* action_dev,
                 const float* reward_dev, const uint8_t* terminal_dev, int32_t batch, int32_t global_batch, uint32_t flags,
                 void* stream);

/* iDQN.update_target_params, T-step (idqn.py:78-80): target <- online (a REAL copy; the reference
 * aliases immutable arrays), then online[k] <- online[k+1] for k < K-1.  Adam state is not shifted. */
int idqn_target_update(idqn_handle_t h, void* stream);
/* D-step, sync_target_params (idqn.py:20-24,92): target[k] <- online[k-1] for k >= 1.              */
int idqn_target_sync(idqn_handle_t h, void* stream);

/* network.apply(params[head], states) (idqn.py:131, dqn.py:90): Q-values of one head for n <= 32
 * states; which = 0 online / 1 target.  q_out_dev: float32 [n][A].                                 */
int idqn_q_values(idqn_handle_t h, int32_t which, int32_t head, const void* states_dev, int32_t n,
                  float* q_out_dev, void* stream);
/* `network.apply(params[idx], state)` followed by `jnp.argmax` (idqn.py:126-131, dqn.py:88-92): the greedy action of
 * each of the n states (first maximum on ties) as int32 in action_out_dev [n]; the Q-values are left in q_out_dev.   */
int idqn_best_action(idqn_handle_t h, int32_t which, int32_t head, const void* states_dev, int32_t n,
                     float* q_out_dev, int32_t* action_out_dev, void* stream);

/* The greedy branch of select_action for ONE state in PINNED host memory (slimdqn/sample_collection/utils.py:8-21:
 * upload, `best_action`, blocking `.item()`): uint8 pixels for the cnn, float32 features for the fc arch.  The action of
 * head `head` is in action_host_pinned[0] when the call returns; q_out_dev receives the A Q-values (device memory,
 * ordered on `stream` like any other result).  By default the last kernel writes the action into a mapped host mailbox
 * that the call polls, so it returns WITHOUT a stream synchronisation (work queued on `stream` behind it may still be
 * running); IDQN_ACT_POLL=0 restores a device-to-host copy plus hipStreamSynchronize.  After the first call per (net,
 * buffers) the whole sequence is replayed as one hipGraph (IDQN_ACT_GRAPH=0: eager).                                   */
int idqn_act_host(idqn_handle_t h, int32_t which, int32_t head, const void* state_host_pinned, float* q_out_dev,
                  int32_t* action_host_pinned, void* stream);
/* The same in two halves, so that host work that does not depend on the action (the replay-buffer bookkeeping of the
 * PREVIOUS transition, key splits) runs while the GPU computes it: _begin uploads and launches and returns at once, _end
 * waits for the action exactly as idqn_act_host does.  One launch may be pending per handle.                         */
int idqn_act_host_begin(idqn_handle_t h, int32_t which, int32_t head, const void* state_host_pinned, float* q_out_dev,
                        int32_t* action_host_pinned, void* stream);
int idqn_act_host_end(idqn_handle_t h, int32_t* action_host_pinned, void* stream);

/* Test / debug access to internal activation buffers by name (device pointer + byte size).        */
int idqn_debug_buffer(idqn_handle_t h, const char* name, void** ptr_dev, int64_t* nbytes);
/* Mean duration (ms) and launch count of the dominant kernel over the IDQN_F_PROFILE calls since
 * the last read; synchronises the events it reads.                                                 */
int idqn_profile_read(idqn_handle_t h, double* mean_ms, int32_t* n_launches, char* kernel_name /*[64]*/);
/* Per-launch table of the IDQN_F_PROFILE_ALL steps since the last read (at most ~100 steps are kept): one line per
 * launch of a step, "name\tmean microseconds\tcount\n", in launch order.  A launch's time is the span between the
 * event behind the previous launch and the event behind it (its duration plus its dispatch gap).                    */
int idqn_profile_table(idqn_handle_t h, char* out, int32_t out_bytes);

/* ------------------------------------------------------------------------------------------
 * Sum tree (slimdqn/sample_collection/sum_tree.py).  nodes_dev: float64 [2**depth - 1] in HBM,
 * depth = ceil(log2(capacity)) + 1, first leaf at 2**(depth-1) - 1 (sum_tree.py:14-17).
 * ---------------------------------------------------------------------------------------- */
/* SumTree.set (sum_tree.py:20-47): deltas against the current leaves BEFORE de-duplication, first
 * occurrence of a duplicate wins, per-level sequential accumulation in ascending node order
 * (bit-exact with np.add.at).  n <= 4096.  scratch_dev: >= 16 * 4096 bytes.                        */
int sumtree_set(double* nodes_dev, int32_t depth, const int32_t* indices_dev, const double* values_dev,
                int32_t n, void* scratch_dev, void* stream);
/* SumTree.get (sum_tree.py:49-51) */
int sumtree_get(const double* nodes_dev, int32_t depth, const int32_t* indices_dev, int32_t n,
                double* out_dev, void* stream);
/* SumTree.query (sum_tree.py:58-102).  status_dev[0] |= 1 if a target is outside [0, root)
 * (-> ValueError, :73-74), |= 2 if the per-level invariant `target < node` fails (-> the reference's
 * assert at :81).  out_dev: int32 leaf indices.                                                     */
int sumtree_query(const double* nodes_dev, int32_t depth, const double* targets_dev, int32_t n,
                  int32_t* out_dev, int32_t* status_dev, void* stream);

/* SumTree.query with ONE host read (sum_tree.py:58-102 returns numpy; samplers.py:105-116 `sample` needs root, leaves and
 * the status of the range check on the host): the n float64 values at values_host are the targets, or -- scale_by_root --
 * the uniforms u_i of the host's PCG64, turned into numpy's `Generator.uniform(0.0, root)` = 0.0 + root * u_i on the device
 * (samplers.py:110).  One launch reads them from a mapped host mailbox, descends, maps leaf -> key through
 * index_to_key_dev (int32 [capacity], NULL: keys = leaves) and writes leaves, keys, root and status back; the host polls
 * the mailbox's sequence number -- no device->host copy, no stream synchronisation, no separate read of the root.
 * status: bit 0 a target outside [0, root) (-> ValueError), bit 1 the per-level assert (:81), bit 2 a leaf >= n_live, the
 * number of valid entries of index_to_key_dev (-> the reference's IndexError from `_index_to_key[index]`, samplers.py:114;
 * its key is returned as -1; n_live < 0: not checked).  root == 0: leaves are 0.  */
int sampler_mailbox_create(int32_t max_n, void** mailbox_out);
int sampler_mailbox_destroy(void* mailbox);
int sumtree_query_host(const double* nodes_dev, int32_t depth, const double* values_host, int32_t n,
                       int32_t scale_by_root, const int32_t* index_to_key_dev, int32_t n_live, void* mailbox,
                       int32_t* leaves_out_host, int32_t* keys_out_host, double* root_out_host,
                       int32_t* status_out_host, void* stream);
/* UniformSamplingDistribution's index -> key map on the device (samplers.py:26-49), for callers that keep sampled keys on
 * the device: add = sampler_map_set(map, len, key); remove = sampler_map_set(map, hole, moved last key) (:31-35: the host
 * map knows both); sample = sampler_map_indices over the int32 indices the host generator drew (:43-49). */
int sampler_map_set(int32_t* index_to_key_dev, int32_t index, int32_t key, void* stream);
int sampler_map_indices(const int32_t* index_to_key_dev, const int32_t* indices_dev, int32_t n, int32_t* keys_out_dev,
                        void* stream);
/* PrioritizedSamplingDistribution.add (samplers.py:62-66): index_to_key[index] = key and tree.set(index, value) for the
 * new last index, ONE launch.  The inverse map key -> index stays a host dict: its only consumers are host-called
 * operations whose arguments are host keys (remove, update). */
int sampler_prioritized_add(double* nodes_dev, int32_t depth, int32_t* index_to_key_dev, int32_t index, int32_t key,
                            double value, void* stream);
/* PrioritizedSamplingDistribution.remove (samplers.py:89-103) for hole = key_to_index[key], last = len - 1: the last
 * entry's priority moves into the hole by the two-leaf set {hole: leaf[last], last: 0.0} (one leaf {hole: 0.0} when
 * hole == last), its key by index_to_key[hole] = index_to_key[last] (:31-35) -- ONE launch, the moved priority is read on
 * the device (the reference reads it with tree.get, :99).  index_to_key_dev may be NULL (tree only). */
int sampler_prioritized_remove(double* nodes_dev, int32_t depth, int32_t* index_to_key_dev, int32_t hole, int32_t last,
                               void* stream);

/* ------------------------------------------------------------------------------------------
 * Prioritized-replay write-back: an EXTENSION with no reference counterpart (the reference's sample() drops the
 * sampled keys, replay_buffer.py:222-230, and its loss has no importance weights, idqn.py:111-112) -- SURVEY 8f-4,
 * "design freely, parity unpinned".  Checked against the oracle's weighted loss / TD errors only.
 * ---------------------------------------------------------------------------------------- */
/* Per-sample loss weights (float [batch], NULL = the reference's plain mean) and an output for |TD error|
 * (float [K][batch], NULL = none) used by every following idqn_learn_on_batch: loss_k = sum_b w_b td_kb^2 / divisor. */
int idqn_set_per_buffers(idqn_handle_t h, const float* weights_dev, float* td_abs_out_dev);
/* SumTree.set for ONE leaf with the index passed by value (no index upload; sum_tree.py:33-47 for n = 1); the new
 * value is `value`, or value_dev[0] when value_dev is not NULL (e.g. the running maximum priority).                  */
int sumtree_set_one(double* nodes_dev, int32_t depth, int32_t index, double value, const double* value_dev,
                    void* stream);
/* Leaves for targets made on the device from uniforms in [0, 1) (float64 [n]): u_i * root, or the stratified
 * (i + u_i) / n * root; same descent as sumtree_query, no host read of the root.                                      */
int per_sample_leaves(const double* nodes_dev, int32_t depth, const double* uniforms_dev, int32_t n,
                      int32_t stratified, int32_t* leaves_out_dev, void* stream);
/* w_i = (n_items * p_i / root)^(-beta) / max_j w_j for the sampled leaves (p from the sum tree).  The leaves are first
 * clamped, in place, to [0, n_items - 1] (a descent can end on an empty leaf through rounding in the sums).            */
int per_importance_weights(const double* nodes_dev, int32_t depth, int32_t* leaves_dev, int32_t n,
                           int64_t n_items, double beta, float* weights_out_dev, void* stream);
/* priority_i = (mean_k or max_k |td[k][i]| + eps)^alpha as float64, ready for sumtree_set on the same leaves;
 * max_priority_dev[0] (may be NULL) keeps the running maximum (sum_tree.py:18,32 `max_recorded_priority`).           */
int per_priorities_from_td(const float* td_abs_dev, int32_t n_heads, int32_t n, int32_t reduce_max, double eps,
                           double alpha, double* priorities_out_dev, double* max_priority_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Replay store in HBM (slimdqn/sample_collection/replay_buffer.py:202-230).  The store the product uses is the
 * frame ring described next (frames written once, elements are 8-int32 rows, slot = key % capacity: keys are the
 * monotonically increasing add_count and eviction is FIFO, :206-213); replay_gather further down serves stores that
 * keep whole (state, next_state) pairs per slot.
 * ---------------------------------------------------------------------------------------- */
/* Frame-ring replay store (the layout ReplayBuffer uses): every environment frame is written to HBM once, at ring
 * slot (transition index % n_frames) of frames_dev [n_frames][frame_elems * itemsize]; a replay element is one row
 * of meta_dev, int32 [capacity][8] = {newest state frame slot, valid state frames, newest next_state frame slot,
 * valid next_state frames, action, reward as f32 bits, terminal, 0}.  This call is ReplayBuffer.sample's fetch +
 * unpack + np.stack (replay_buffer.py:223-229) fused with the accumulator's stack building (:119-137,
 * `state[..., ch] = observation`, zero frames before the episode start): for sample i it writes
 * state_out[i][pixel][ch] = frame[newest - (stack-1-ch)][pixel] (0 where fewer than stack frames are valid), the same
 * for next_state, and the three scalars.  Outputs: [n][frame_elems][stack] elements of `itemsize` bytes.            */
int replay_gather_stacked(const uint8_t* frames_dev, int64_t n_frames, int64_t frame_elems, int32_t itemsize,
                          int32_t stack, const int32_t* meta_dev, const int32_t* slots_dev, int32_t n,
                          uint8_t* state_out_dev, uint8_t* next_state_out_dev, int32_t* action_out_dev,
                          float* reward_out_dev, uint8_t* terminal_out_dev, void* stream);
/* Plain row gather of a store that keeps whole (state, next_state) pairs per slot, [capacity][2][obs_bytes]:
 * copies the sampled slots'
 * state / next_state into contiguous [n][obs_bytes] batches.                                       */
int replay_gather(const uint8_t* store_dev, int64_t obs_bytes, const int32_t* slots_dev, int32_t n,
                  uint8_t* state_out_dev, uint8_t* next_state_out_dev, void* stream);
/* Gather of the per-slot scalars kept as arrays [capacity]: action i32, reward f32, terminal u8.   */
int replay_gather_scalars(const int32_t* action_store_dev, const float* reward_store_dev,
                          const uint8_t* terminal_store_dev, const int32_t* slots_dev, int32_t n,
                          int32_t* action_out_dev, float* reward_out_dev, uint8_t* terminal_out_dev,
                          void* stream);
/* ReplayBuffer.add, device half (replay_buffer.py:206-213 keeps transitions on the host): the newest frame, staged in
 * PINNED host memory, is copied asynchronously into slot `slot` of the frame ring in HBM.  The staging slot may be
 * reused once the stream has passed this copy.  The host mirror also sends the new element rows and the sampled slot
 * indices this way (slot 0 of a byte range: frame_ring_dev = destination, frame_bytes = length).                      */
int replay_add_frame(void* frame_ring_dev, int64_t slot, int64_t frame_bytes, const void* frame_host_pinned, void* stream);
/* Growth of the frame ring (no reference counterpart: the reference's host dict grows by itself,
 * replay_buffer.py:206-213): the live frames, transition indices [first_t, first_t + count), are copied from slot
 * t % old_n of the old ring to slot t % new_n of the new one (count <= old_n <= new_n, distinct buffers).             */
int replay_ring_regrow(const void* old_ring_dev, int64_t old_n, void* new_ring_dev, int64_t new_n, int64_t first_t,
                       int64_t count, int64_t frame_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IDQN_HIP_H */
