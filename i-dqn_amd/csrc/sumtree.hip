// HBM-resident fp64 sum tree: set / get / query, bit-exact with the reference's numpy code.
//
// Replaces slimdqn/sample_collection/sum_tree.py:20-102 of the reference.  The one thing that makes
// `set` non-trivial on a GPU is the accumulation ORDER: the reference sorts the updated leaves
// (np.unique) and then does one np.add.at per tree level, i.e. every node that several updated
// leaves share receives ((node + d0) + d1) + ... strictly in ascending-leaf order.  fp64 addition is
// not associative, so unordered atomics would drift from the reference by ulps.  Here a single
// workgroup sorts (leaf, position) keys in LDS; at each level the first lane of every run of equal
// node indices adds its run's deltas sequentially in a register and stores once.  Runs at one level
// touch distinct nodes and different levels touch distinct nodes, so no memory ordering is needed
// between lanes -- only the LDS arrays are shared.  Latency-bound pointer chasing over a 16.8 MB
// array (capacity 2^20); HBM bytes are negligible (n * depth * 16 B).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";
void idqn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* idqn_last_error(void) { return g_err; }
extern "C" int idqn_abi_version(void) { return 4; }  // 2: idqn_config_t.n_quantiles, the i-IQN entry points; 3: sumtree_query_host(n_live), the idqn_dp_* family; 4: idqn_learn_on_replay

#define ST_THREADS 1024
#define ST_MAX_N 4096

// keys: (node index << 32) | original position; padded with ~0.
__global__ __launch_bounds__(ST_THREADS) void k_sumtree_set(double* __restrict__ nodes, int depth,
                                                            const int32_t* __restrict__ idx,
                                                            const double* __restrict__ val, int n, int m,
                                                            double* __restrict__ delta_scratch) {
    __shared__ unsigned long long key[ST_MAX_N];
    __shared__ unsigned int cur[ST_MAX_N];
    const int tid = threadIdx.x;
    const unsigned int first_leaf = (1u << (depth - 1)) - 1u;
#ifdef ST_PROF
#define ST_STAMP(i) if (tid == 0) reinterpret_cast<long long*>(delta_scratch)[6000 + (i)] = wall_clock64();
#else
#define ST_STAMP(i)
#endif
    ST_STAMP(0)
    // 1. deltas against the CURRENT leaf values, before any de-duplication (sum_tree.py:33-34)
    for (int i = tid; i < m; i += ST_THREADS) {
        if (i < n) {
            unsigned int leaf = first_leaf + (unsigned int)idx[i];
            delta_scratch[i] = val[i] - nodes[leaf];
            key[i] = ((unsigned long long)leaf << 32) | (unsigned int)i;
        } else {
            key[i] = ~0ull;
        }
    }
    __syncthreads();
    ST_STAMP(1)
    // 2. sort by (leaf, position): ascending leaves, first occurrence first (np.unique).  Minibatch-sized sets sort by rank
    //    counting (keys are unique; LDS broadcast reads, four barriers in all) instead of the bitonic network's 36-45.
    if (m <= ST_THREADS / 2) {  // (n^2 comparisons: past 512 keys the bitonic network's 55 barriers are cheaper)
        // P threads per key (P = 1024 / m, a power of two): thread (key i = tid % m, part = tid / m) counts the keys below
        // key i among every P-th element; the partial counts meet in an LDS integer (order-free).  With one thread per
        // key the 256-leaf write-back spent 13.8 us here: 12 of the 16 waves had nothing to count but still walked the loop.
        const int P = ST_THREADS / m, i = tid & (m - 1), part = tid / m;
        const unsigned long long mine = i < n ? key[i] : ~0ull;
        if (tid < m) cur[tid] = 0u;  // (cur is free until phase 3)
        __syncthreads();
        if (i < n) {
            int rank = 0;
            for (int j0 = part; j0 < n; j0 += 8 * P) {
                unsigned long long kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kk[u] = key[min(j0 + u * P, n - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += (j0 + u * P < n && kk[u] < mine) ? 1 : 0;
            }
            atomicAdd(&cur[i], (unsigned int)rank);
        }
        __syncthreads();
        const unsigned int rk = tid < n ? cur[tid] : 0u;
        const unsigned long long mine0 = tid < n ? key[tid] : ~0ull;
        __syncthreads();
        if (tid < n) key[rk] = mine0;
        __syncthreads();
    } else
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += ST_THREADS) {
                int p = i ^ j;
                if (p > i) {
                    unsigned long long a = key[i], b = key[p];
                    bool asc = (i & k) == 0;
                    if ((a > b) == asc) {
                        key[i] = b;
                        key[p] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    ST_STAMP(2)
    // 3. sorted deltas; duplicates of a leaf (every occurrence but the first) contribute exactly +0.0
    double dl[ST_MAX_N / ST_THREADS];
#pragma unroll
    for (int q = 0; q < ST_MAX_N / ST_THREADS; ++q) {
        int s = tid + q * ST_THREADS;
        dl[q] = 0.0;
        if (s < n) {
            unsigned int leaf = (unsigned int)(key[s] >> 32);
            bool dup = s > 0 && (unsigned int)(key[s - 1] >> 32) == leaf;
            dl[q] = dup ? 0.0 : delta_scratch[(unsigned int)key[s]];
            cur[s] = leaf;
        }
    }
    __syncthreads();
    // re-use the key array (as doubles) for the sorted deltas
    double* sdelta = reinterpret_cast<double*>(key);
#pragma unroll
    for (int q = 0; q < ST_MAX_N / ST_THREADS; ++q) {
        int s = tid + q * ST_THREADS;
        if (s < n) sdelta[s] = dl[q];
    }
    __syncthreads();
    // 4. every (level, run of equal ancestors) pair at once: a node belongs to exactly one level, so the levels are
    //    independent and all their read-modify-writes are in flight together (one memory round trip instead of
    //    `depth` dependent ones); the head of each run accumulates its run in ascending leaf order, which is the
    //    order np.add.at applies the sorted deltas in (sum_tree.py:39-47).  cur[] holds the sorted leaf nodes; the
    //    ancestor `level` levels up of 0-based heap node x is ((x + 1) >> level) - 1.
    ST_STAMP(3)
    const int pairs = n * depth;
    // Eight (leaf, level) pairs per thread and round: the node reads of all eight are in flight together (one memory round
    // trip per round instead of one per pair -- a 256-leaf write-back has 5376 pairs, i.e. 5-6 per thread).
    for (int pr0 = tid; pr0 < pairs; pr0 += 8 * ST_THREADS) {
        double xs[8];
        unsigned int nd[8];
        int st[8], lv[8];
        bool head[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pr = pr0 + u * ST_THREADS;
            head[u] = false;
            if (pr < pairs) {
                const int s0 = pr / depth, level = pr - s0 * depth;  // the long runs near the root land on different lanes
                const unsigned int node = ((cur[s0] + 1u) >> level) - 1u;
                st[u] = s0; lv[u] = level; nd[u] = node;
                head[u] = s0 == 0 || ((cur[s0 - 1] + 1u) >> level) - 1u != node;
                if (head[u]) xs[u] = nodes[node];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!head[u]) continue;
            // The run's deltas are added strictly one after the other (that order IS the result).  The end of the run does
            // not depend on the running sum: it is found first (ancestors of sorted leaves never decrease: a binary search,
            // skipped for the common run of one), so the dependent chain is ONE fp64 add per element -- with the end test
            // inside the chain the root's run cost ~100 cycles per element (13.8 us of a 256-leaf write-back).
            const unsigned int node = nd[u];
            const int level = lv[u];
            double x = xs[u];
            int e = st[u], end = e + 1;
            if (end < n && ((cur[end] + 1u) >> level) - 1u == node) {
                int lo = end, hi = n;  // invariant: elements [st, lo] belong to the run, element hi does not (or hi == n)
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (((cur[mid] + 1u) >> level) - 1u == node) lo = mid; else hi = mid;
                }
                end = hi;
            }
            for (; e + 8 <= end; e += 8) {
                double d[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) d[w] = sdelta[e + w];
#pragma unroll
                for (int w = 0; w < 8; ++w) x = x + d[w];
            }
            for (; e < end; ++e) x = x + sdelta[e];
            nodes[node] = x;
        }
    }
    ST_STAMP(4)
}

__global__ void k_sumtree_get(const double* __restrict__ nodes, int depth, const int32_t* __restrict__ idx, int n,
                              double* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = nodes[((1u << (depth - 1)) - 1u) + (unsigned int)idx[i]];
}

// Level-synchronous descent, one lane per target (sum_tree.py:76-102).
__global__ void k_sumtree_query(const double* __restrict__ nodes, int depth, const double* __restrict__ targets,
                                int n, int32_t* __restrict__ out, int32_t* __restrict__ status) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int first_leaf = (1u << (depth - 1)) - 1u;
    double t = targets[i];
    int bad = 0;
    if (!(t >= 0.0 && t < nodes[0])) bad |= 1;  // ValueError in the reference (:73-74)
    unsigned int node = 0;
    while (node < first_leaf) {
        if (!(t < nodes[node])) bad |= 2;  // the reference's per-level assert (:81)
        unsigned int left = 2u * node + 1u;
        double ls = nodes[left];
        if (t < ls) {
            node = left;
        } else {
            t = t - ls;
            node = left + 1u;
        }
    }
    out[i] = (int32_t)(node - first_leaf);
    if (bad) atomicOr(status, bad);
}

// Latency-oriented variant for minibatch-sized queries: ONE WAVE PER QUERY.  The 2^(s+1)-1 nodes of the s <= 5
// levels below the current node are fetched by the 64 lanes in one round trip, then the wave descends those levels
// out of registers (shuffles): depth 21 costs 4 dependent memory round trips instead of 20.  Same comparisons and
// the same `t -= left_sum` sequence as the scalar walk, so the result is bit-identical.
__device__ __forceinline__ unsigned int wave_descend(const double* __restrict__ nodes, int depth, double t, int& bad) {
    const int lane = threadIdx.x & 63;
    unsigned int node = 0;
    int level = 0;
    const int h = lane + 1;                 // 1-based heap number inside the sub-tree; lane 63 idles
    const int j = 31 - __clz(h);            // its level inside the sub-tree
    const unsigned int p = h - (1u << j);  // its position in that level
    while (level < depth - 1) {
        const int s = min(5, depth - 1 - level);
        double v = 0.0;
        if (j <= s && lane < 63) v = nodes[(size_t)(node + 1u) * (1u << j) + p - 1u];
        int cur = 1;
        for (int step = 0; step < s; ++step) {
            const double here = __shfl(v, cur - 1);
            const double ls = __shfl(v, 2 * cur - 1);
            if (!(t < here)) bad |= 2;
            if (t < ls) {
                cur = 2 * cur;
            } else {
                t = t - ls;
                cur = 2 * cur + 1;
            }
        }
        const int jj = 31 - __clz(cur);
        node = (node + 1u) * (1u << jj) + (cur - (1u << jj)) - 1u;
        level += s;
    }
    return node;
}

__global__ __launch_bounds__(256) void k_sumtree_query_wave(const double* __restrict__ nodes, int depth,
                                                            const double* __restrict__ targets, int n,
                                                            int32_t* __restrict__ out, int32_t* __restrict__ status) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;  // wave-uniform
    const unsigned int first_leaf = (1u << (depth - 1)) - 1u;
    const double t = targets[i];
    int bad = 0;
    if (!(t >= 0.0 && t < nodes[0])) bad |= 1;
    const unsigned int node = wave_descend(nodes, depth, t, bad);
    if ((threadIdx.x & 63) == 0) {
        out[i] = (int32_t)(node - first_leaf);
        if (bad) atomicOr(status, bad);
    }
}

// Extension (prioritized-replay loop without a host round trip for the root): targets are made on the device from
// uniforms in [0, 1): u * root, or the stratified (i + u) / n * root, clamped below the root.
__global__ __launch_bounds__(256) void k_per_sample(const double* __restrict__ nodes, int depth,
                                                    const double* __restrict__ uniforms, int n, int stratified,
                                                    int32_t* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const unsigned int first_leaf = (1u << (depth - 1)) - 1u;
    const double root = nodes[0];
    double t = stratified ? ((double)i + uniforms[i]) / (double)n * root : uniforms[i] * root;
    t = fmin(t, nextafter(root, 0.0));
    int bad = 0;
    const unsigned int node = (root > 0.0) ? wave_descend(nodes, depth, t, bad) : first_leaf;
    if ((threadIdx.x & 63) == 0) out[i] = (int32_t)(node - first_leaf);
}

extern "C" int sumtree_set(double* nodes_dev, int32_t depth, const int32_t* indices_dev, const double* values_dev,
                           int32_t n, void* scratch_dev, void* stream) {
    IDQN_REQUIRE(nodes_dev && indices_dev && values_dev && scratch_dev, "sumtree_set: null pointer");
    IDQN_REQUIRE(depth >= 1 && depth <= 31, "sumtree_set: depth %d out of range", depth);
    IDQN_REQUIRE(n >= 1 && n <= ST_MAX_N, "sumtree_set: n = %d, must be in [1, %d]", n, ST_MAX_N);
    int m = 1;
    while (m < n) m <<= 1;
    hipLaunchKernelGGL(k_sumtree_set, dim3(1), dim3(ST_THREADS), 0, (hipStream_t)stream, nodes_dev, depth, indices_dev,
                       values_dev, n, m, (double*)scratch_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int sumtree_get(const double* nodes_dev, int32_t depth, const int32_t* indices_dev, int32_t n,
                           double* out_dev, void* stream) {
    IDQN_REQUIRE(nodes_dev && indices_dev && out_dev && n >= 1, "sumtree_get: bad arguments");
    hipLaunchKernelGGL(k_sumtree_get, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, nodes_dev, depth,
                       indices_dev, n, out_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int sumtree_query(const double* nodes_dev, int32_t depth, const double* targets_dev, int32_t n,
                             int32_t* out_dev, int32_t* status_dev, void* stream) {
    IDQN_REQUIRE(nodes_dev && targets_dev && out_dev && status_dev && n >= 1, "sumtree_query: bad arguments");
    IDQN_REQUIRE(depth >= 1 && depth <= 31, "sumtree_query: depth %d out of range", depth);
    if (n <= 2048)  // minibatch-sized: latency matters, one wave per query
        hipLaunchKernelGGL(k_sumtree_query_wave, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, nodes_dev, depth,
                           targets_dev, n, out_dev, status_dev);
    else
        hipLaunchKernelGGL(k_sumtree_query, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, nodes_dev, depth,
                           targets_dev, n, out_dev, status_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

// ---------------------------------------------------------------------------------------------------
// Prioritized-replay EXTENSION (the reference has no write-back path: its sample() drops the keys,
// replay_buffer.py:222-230, so there is no behaviour to match here -- parity unpinned).
// ---------------------------------------------------------------------------------------------------
// w_i = (n_items * p_i / root)^(-beta), normalised by the largest weight of the batch (Schaul et al. 2016, eq. 2).
__global__ __launch_bounds__(1024) void k_per_weights(const double* __restrict__ nodes, int depth,
                                                      int32_t* __restrict__ leaves, int n, double n_items,
                                                      double beta, float* __restrict__ out) {
    __shared__ double red[1024];
    const unsigned int first_leaf = (1u << (depth - 1)) - 1u;
    const double root = nodes[0];
    double wmax = 0.0;
    // a descent can land on an empty leaf at or past the item count when rounding in the tree sums leaves a sliver of
    // mass there: pull it back onto the last live leaf, so that the gather that follows stays inside the store
    for (int i = threadIdx.x; i < n; i += 1024) leaves[i] = min(max(leaves[i], 0), (int32_t)n_items - 1);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double p = nodes[first_leaf + (unsigned int)leaves[i]];
        const double w = (p > 0.0 && root > 0.0) ? pow(n_items * p / root, -beta) : 0.0;
        wmax = fmax(wmax, w);
    }
    red[threadIdx.x] = wmax;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    wmax = red[0];
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double p = nodes[first_leaf + (unsigned int)leaves[i]];
        const double w = (p > 0.0 && root > 0.0) ? pow(n_items * p / root, -beta) : 0.0;
        out[i] = wmax > 0.0 ? (float)(w / wmax) : 1.0f;
    }
}

// priority_i = (reduce_k |td[k][i]| + eps)^alpha ; reduce = mean (0) or max (1) over the K heads; also keeps the
// running maximum priority (the reference's `max_recorded_priority`, sum_tree.py:18,32) in max_dev[0].
__global__ void k_per_priorities(const float* __restrict__ td_abs, int K, int n, int reduce_max, double eps,
                                 double alpha, double* __restrict__ out, double* __restrict__ max_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
        const double v = (double)td_abs[(long)k * n + i];
        acc = reduce_max ? fmax(acc, v) : acc + v;
    }
    if (!reduce_max) acc /= (double)K;
    const double pr = pow(acc + eps, alpha);
    out[i] = pr;
    if (max_dev) atomicMax(reinterpret_cast<unsigned long long*>(max_dev), (unsigned long long)__double_as_longlong(pr));
}

// One leaf, index passed by value (no index upload): leaf <- value, every ancestor += (value - old leaf) -- exactly
// what SumTree.set does for a single element (sum_tree.py:33-47).  value_dev, when given, overrides `value`.
// One wave: lane l owns the ancestor l levels up (depth <= 31 nodes, all distinct), so the read-modify-writes of the
// whole path are in flight together -- two dependent memory round trips instead of `depth` (the single-thread walk took
// 23 us on a 2^20-leaf tree).  Same arithmetic per node: node + delta.
__global__ __launch_bounds__(64) void k_sumtree_set_one(double* __restrict__ nodes, int depth, int index, double value,
                                                       const double* __restrict__ value_dev, int32_t* __restrict__ index_to_key,
                                                       int key) {
    const unsigned int leaf = ((1u << (depth - 1)) - 1u) + (unsigned int)index;
    const int lane = threadIdx.x;
    const double v = value_dev ? value_dev[0] : value;
    const double delta = v - nodes[leaf];  // (every lane reads the same word: one request)
    if (lane < depth) {
        const unsigned int node = ((leaf + 1u) >> lane) - 1u;
        nodes[node] = nodes[node] + delta;
    }
    if (index_to_key && lane == 0) index_to_key[index] = key;  // samplers.py:64-65 (PrioritizedSamplingDistribution.add)
}

// PrioritizedSamplingDistribution.remove (samplers.py:89-103) without the host read of the moved priority: the last
// entry's priority v = leaf[last] goes into the hole by the two-leaf set {hole: v, last: 0.0} (np.unique order: hole <
// last; deltas against the current leaves; where the two paths meet the node accumulates (x + d_hole) + d_last, as
// np.add.at does), and its key moves in index_to_key.  hole == last: the one-leaf set {hole: 0.0}.
// One wave, lane = level (all nodes of the two paths are distinct per level).
__global__ __launch_bounds__(64) void k_sampler_remove(double* __restrict__ nodes, int depth, int32_t* __restrict__ index_to_key,
                                                      int hole, int last) {
    const unsigned int first_leaf = (1u << (depth - 1)) - 1u;
    const unsigned int lh = first_leaf + (unsigned int)hole, ll = first_leaf + (unsigned int)last;
    const int lane = threadIdx.x;
    const double v = nodes[ll];
    const double d_hole = (hole == last ? 0.0 : v) - nodes[lh];
    const double d_last = 0.0 - v;
    const int moved = index_to_key ? index_to_key[last] : 0;
    if (lane < depth) {
        const unsigned int nh = ((lh + 1u) >> lane) - 1u, nl = ((ll + 1u) >> lane) - 1u;
        if (hole == last) {
            nodes[nh] = nodes[nh] + d_hole;
        } else if (nh == nl) {
            nodes[nh] = (nodes[nh] + d_hole) + d_last;
        } else {
            const double a = nodes[nh], b = nodes[nl];
            nodes[nh] = a + d_hole;
            nodes[nl] = b + d_last;
        }
    }
    if (index_to_key && lane == 0 && hole != last) index_to_key[hole] = moved;  // samplers.py:31-35 swap-with-last
}

extern "C" int sumtree_set_one(double* nodes_dev, int32_t depth, int32_t index, double value, const double* value_dev,
                               void* stream) {
    IDQN_REQUIRE(nodes_dev && depth >= 1 && depth <= 31, "sumtree_set_one: bad arguments");
    IDQN_REQUIRE(index >= 0 && (int64_t)index < ((int64_t)1 << (depth - 1)), "sumtree_set_one: index %d out of range", index);
    IDQN_REQUIRE(value_dev || value >= 0.0, "sumtree_set_one: negative value");
    hipLaunchKernelGGL(k_sumtree_set_one, dim3(1), dim3(64), 0, (hipStream_t)stream, nodes_dev, depth, index, value, value_dev,
                       (int32_t*)nullptr, 0);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int sampler_prioritized_add(double* nodes_dev, int32_t depth, int32_t* index_to_key_dev, int32_t index, int32_t key,
                                       double value, void* stream) {
    IDQN_REQUIRE(nodes_dev && index_to_key_dev && depth >= 1 && depth <= 31, "sampler_prioritized_add: bad arguments");
    IDQN_REQUIRE(index >= 0 && (int64_t)index < ((int64_t)1 << (depth - 1)), "sampler_prioritized_add: index %d out of range", index);
    IDQN_REQUIRE(value >= 0.0, "sampler_prioritized_add: negative value");
    hipLaunchKernelGGL(k_sumtree_set_one, dim3(1), dim3(64), 0, (hipStream_t)stream, nodes_dev, depth, index, value,
                       (const double*)nullptr, index_to_key_dev, key);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int sampler_prioritized_remove(double* nodes_dev, int32_t depth, int32_t* index_to_key_dev, int32_t hole, int32_t last,
                                          void* stream) {
    IDQN_REQUIRE(nodes_dev && depth >= 1 && depth <= 31, "sampler_prioritized_remove: bad arguments");
    IDQN_REQUIRE(hole >= 0 && hole <= last && (int64_t)last < ((int64_t)1 << (depth - 1)),
                 "sampler_prioritized_remove: hole %d / last %d out of range", hole, last);
    hipLaunchKernelGGL(k_sampler_remove, dim3(1), dim3(64), 0, (hipStream_t)stream, nodes_dev, depth, index_to_key_dev, hole, last);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

// UniformSamplingDistribution's map on the device (samplers.py:26-49): add writes index_to_key[len] = key, remove writes the
// moved last key into the hole -- both are ONE int32 store, passed by value (the host map knows index and key); sample maps
// the host generator's indices to keys with a gather.  Only a sampler that opts in pays for it (the reference protocol hands
// host keys to host code; the device copy serves callers that keep the sampled keys on the device).
__global__ void k_sampler_map_set(int32_t* __restrict__ index_to_key, int index, int key) { index_to_key[index] = key; }
__global__ void k_sampler_map_indices(const int32_t* __restrict__ index_to_key, const int32_t* __restrict__ indices, int n,
                                      int32_t* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = index_to_key[indices[i]];
}
extern "C" int sampler_map_set(int32_t* index_to_key_dev, int32_t index, int32_t key, void* stream) {
    IDQN_REQUIRE(index_to_key_dev && index >= 0, "sampler_map_set: bad arguments");
    hipLaunchKernelGGL(k_sampler_map_set, dim3(1), dim3(1), 0, (hipStream_t)stream, index_to_key_dev, index, key);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}
extern "C" int sampler_map_indices(const int32_t* index_to_key_dev, const int32_t* indices_dev, int32_t n, int32_t* keys_out_dev,
                                   void* stream) {
    IDQN_REQUIRE(index_to_key_dev && indices_dev && keys_out_dev && n >= 1, "sampler_map_indices: bad arguments");
    hipLaunchKernelGGL(k_sampler_map_indices, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, index_to_key_dev, indices_dev, n,
                       keys_out_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

// ---------------------------------------------------------------------------------------------------
// One host read per query / sample (reference protocol: SumTree.query returns numpy, sum_tree.py:58-102;
// PrioritizedSamplingDistribution.sample returns host keys, samplers.py:105-116).  The n targets (or the n uniforms the
// host's PCG64 produced) sit in a mapped, coherent host mailbox; ONE launch reads them from there, descends, maps the
// leaves through index_to_key and writes leaves, keys, the root and the status bits back into the mailbox; the last wave
// to finish (an arrival counter in device memory) writes the sequence number the host polls.  No device -> host copy,
// no stream synchronisation, no separate read of the root.
// ---------------------------------------------------------------------------------------------------
struct SamplerMailbox {
    unsigned char* host;  // mapped + coherent: [0] f64 root, [8] i32 status, [12] u32 seq, [64] f64 in[max_n], i32 leaves[max_n], i32 keys[max_n]
    unsigned char* dev;   // its device address
    unsigned* ctl;        // device: [0] arrivals, [1] status bits of the launch in flight
    unsigned seq;
    int max_n;
};

__global__ __launch_bounds__(256) void k_sumtree_query_mail(const double* __restrict__ nodes, int depth, int n, int scale_by_root,
                                                            const int32_t* __restrict__ index_to_key, int n_live, unsigned char* mail,
                                                            int max_n, unsigned* ctl, unsigned seq) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;  // wave-uniform
    const unsigned int first_leaf = (1u << (depth - 1)) - 1u;
    const double* in = reinterpret_cast<const double*>(mail + 64);
    int32_t* leaves = reinterpret_cast<int32_t*>(mail + 64 + (size_t)max_n * 8);
    int32_t* keys = leaves + max_n;
    const double root = nodes[0];
    // numpy's Generator.uniform(0.0, root): low + (high - low) * next_double  (sum_tree targets of samplers.py:110)
    const double t = scale_by_root ? 0.0 + (root - 0.0) * in[i] : in[i];
    int bad = 0;
    if (!(t >= 0.0 && t < root)) bad |= 1;  // ValueError in the reference (sum_tree.py:73-74)
    unsigned int node = first_leaf;
    if (root > 0.0) node = wave_descend(nodes, depth, t, bad);
    if ((threadIdx.x & 63) == 0) {
        const int leaf = (int)(node - first_leaf);
        leaves[i] = leaf;
        // the map holds n_live entries: a descent that ends on an empty leaf behind them (drift in the node sums, a target at
        // the root) is the reference's IndexError from `_index_to_key[index]` (samplers.py:114), not a stale or foreign key
        const bool past = index_to_key && n_live >= 0 && leaf >= n_live;
        if (past) bad |= 4;
        keys[i] = index_to_key ? (past ? -1 : index_to_key[leaf]) : leaf;
        if (bad) atomicOr(&ctl[1], (unsigned)bad);
        __threadfence_system();  // this wave's results are visible to the host before its arrival counts
        if (atomicAdd(&ctl[0], 1u) == (unsigned)n - 1u) {  // the last wave of the launch announces it
            const unsigned st = atomicExch(&ctl[1], 0u);
            ctl[0] = 0u;
            *reinterpret_cast<double*>(mail) = root;
            *reinterpret_cast<volatile int32_t*>(mail + 8) = (int32_t)st;
            __threadfence_system();
            *reinterpret_cast<volatile unsigned*>(mail + 12) = seq;
        }
    }
}

extern "C" int sampler_mailbox_create(int32_t max_n, void** mailbox_out) {
    IDQN_REQUIRE(mailbox_out && max_n >= 1 && max_n <= (1 << 20), "sampler_mailbox_create: bad arguments");
    SamplerMailbox* mb = new SamplerMailbox();
    mb->max_n = max_n; mb->seq = 0;
    const size_t bytes = 64 + (size_t)max_n * 16;
    hipError_t e = hipHostMalloc((void**)&mb->host, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) { memset(mb->host, 0, bytes); e = hipHostGetDevicePointer((void**)&mb->dev, mb->host, 0); }
    if (e == hipSuccess) e = hipMalloc((void**)&mb->ctl, 64);
    if (e == hipSuccess) e = hipMemset(mb->ctl, 0, 64);
    // hipMemset of device memory returns before it has run, and the null stream does not order against the (non-blocking)
    // stream of the first query: on an idle stream of its own that launch started before the counters were zero
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        if (mb->ctl) (void)hipFree(mb->ctl);
        if (mb->host) (void)hipHostFree(mb->host);
        delete mb;
        IDQN_HIP_CHECK(e);
    }
    *mailbox_out = mb;
    return IDQN_OK;
}

extern "C" int sampler_mailbox_destroy(void* mailbox) {
    SamplerMailbox* mb = (SamplerMailbox*)mailbox;
    if (!mb) return IDQN_OK;
    (void)hipFree(mb->ctl);
    (void)hipHostFree(mb->host);
    delete mb;
    return IDQN_OK;
}

extern "C" int sumtree_query_host(const double* nodes_dev, int32_t depth, const double* values_host, int32_t n,
                                  int32_t scale_by_root, const int32_t* index_to_key_dev, int32_t n_live, void* mailbox,
                                  int32_t* leaves_out_host, int32_t* keys_out_host, double* root_out_host,
                                  int32_t* status_out_host, void* stream) {
    SamplerMailbox* mb = (SamplerMailbox*)mailbox;
    IDQN_REQUIRE(nodes_dev && values_host && mb && leaves_out_host && root_out_host && status_out_host, "sumtree_query_host: null pointer");
    IDQN_REQUIRE(depth >= 1 && depth <= 31, "sumtree_query_host: depth %d out of range", depth);
    IDQN_REQUIRE(n >= 1 && n <= mb->max_n, "sumtree_query_host: n = %d, the mailbox holds %d", n, mb->max_n);
    hipStream_t q = (hipStream_t)stream;
    memcpy(mb->host + 64, values_host, (size_t)n * 8);
    const unsigned want = ++mb->seq;
    hipLaunchKernelGGL(k_sumtree_query_mail, dim3(cdiv(n, 4)), dim3(256), 0, q, nodes_dev, depth, n, scale_by_root, index_to_key_dev,
                       (int)n_live, mb->dev, mb->max_n, mb->ctl, want);
    IDQN_HIP_CHECK(hipGetLastError());
    volatile unsigned* seqp = reinterpret_cast<volatile unsigned*>(mb->host + 12);
    bool seen = false;
    hipError_t qe = hipSuccess;
    for (long spin = 0; spin < (1L << 34); ++spin) {  // (far longer than anything queued in front of the launch)
        if (*seqp == want) { seen = true; break; }
        __builtin_ia32_pause();
        if ((spin & 0xfffff) == 0xfffff && (qe = hipStreamQuery(q)) != hipErrorNotReady) {
            // the stream has drained (or reports an error): the results were written before the kernel ended
            (void)hipStreamSynchronize(q);
            seen = *seqp == want;
            break;
        }
    }
    if (!seen) {
        (void)hipStreamSynchronize(q);
        unsigned ctl_now[2] = {0, 0};
        (void)hipMemcpy(ctl_now, mb->ctl, 8, hipMemcpyDeviceToHost);
        (void)hipMemset(mb->ctl, 0, 64);
        (void)hipStreamSynchronize(nullptr);
        IDQN_REQUIRE(false, "sumtree_query_host: the launch finished without delivering its results (stream query: %s, mailbox seq %u, wanted %u, "
                            "arrivals %u of %d)", hipGetErrorName(qe), *seqp, want, ctl_now[0], n);
    }
    *root_out_host = *reinterpret_cast<const double*>(mb->host);
    *status_out_host = *reinterpret_cast<const int32_t*>(mb->host + 8);
    const int32_t* lv = reinterpret_cast<const int32_t*>(mb->host + 64 + (size_t)mb->max_n * 8);
    memcpy(leaves_out_host, lv, (size_t)n * 4);
    if (keys_out_host) memcpy(keys_out_host, lv + mb->max_n, (size_t)n * 4);
    return IDQN_OK;
}

extern "C" int per_sample_leaves(const double* nodes_dev, int32_t depth, const double* uniforms_dev, int32_t n,
                                 int32_t stratified, int32_t* leaves_out_dev, void* stream) {
    IDQN_REQUIRE(nodes_dev && uniforms_dev && leaves_out_dev && n >= 1, "per_sample_leaves: bad arguments");
    IDQN_REQUIRE(depth >= 1 && depth <= 31, "per_sample_leaves: depth %d out of range", depth);
    hipLaunchKernelGGL(k_per_sample, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, nodes_dev, depth, uniforms_dev, n,
                       stratified, leaves_out_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int per_importance_weights(const double* nodes_dev, int32_t depth, int32_t* leaves_dev, int32_t n,
                                      int64_t n_items, double beta, float* weights_out_dev, void* stream) {
    IDQN_REQUIRE(nodes_dev && leaves_dev && weights_out_dev && n >= 1 && n_items >= 1, "per_importance_weights: bad arguments");
    IDQN_REQUIRE(depth >= 1 && depth <= 31, "per_importance_weights: depth %d out of range", depth);
    hipLaunchKernelGGL(k_per_weights, dim3(1), dim3(1024), 0, (hipStream_t)stream, nodes_dev, depth, leaves_dev, n,
                       (double)n_items, beta, weights_out_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int per_priorities_from_td(const float* td_abs_dev, int32_t n_heads, int32_t n, int32_t reduce_max, double eps,
                                      double alpha, double* priorities_out_dev, double* max_priority_dev, void* stream) {
    IDQN_REQUIRE(td_abs_dev && priorities_out_dev && n_heads >= 1 && n >= 1, "per_priorities_from_td: bad arguments");
    hipLaunchKernelGGL(k_per_priorities, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, td_abs_dev, n_heads, n,
                       reduce_max, eps, alpha, priorities_out_dev, max_priority_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}
