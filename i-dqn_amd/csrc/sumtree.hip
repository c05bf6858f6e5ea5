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
extern "C" int idqn_abi_version(void) { return 1; }

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
    // 2. bitonic sort by (leaf, position): ascending leaves, first occurrence first (np.unique)
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
    const int pairs = n * depth;
    for (int pr = tid; pr < pairs; pr += ST_THREADS) {
        const int s0 = pr / depth, level = pr - s0 * depth;  // the long runs near the root land on different lanes
        const unsigned int node = ((cur[s0] + 1u) >> level) - 1u;
        if (s0 == 0 || ((cur[s0 - 1] + 1u) >> level) - 1u != node) {
            double x = nodes[node];
            int e = s0;
            do {
                x = x + sdelta[e];
                ++e;
            } while (e < n && ((cur[e] + 1u) >> level) - 1u == node);
            nodes[node] = x;
        }
    }
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
__global__ __launch_bounds__(256) void k_sumtree_query_wave(const double* __restrict__ nodes, int depth,
                                                            const double* __restrict__ targets, int n,
                                                            int32_t* __restrict__ out, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;  // wave-uniform
    const unsigned int first_leaf = (1u << (depth - 1)) - 1u;
    double t = targets[i];
    int bad = 0;
    if (!(t >= 0.0 && t < nodes[0])) bad |= 1;
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
    if (lane == 0) {
        out[i] = (int32_t)(node - first_leaf);
        if (bad) atomicOr(status, bad);
    }
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
