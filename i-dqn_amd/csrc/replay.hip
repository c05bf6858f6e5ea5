// HBM replay store: gather of sampled slots into a contiguous minibatch.
//
// Replaces the fetch + unpack + np.stack of ReplayBuffer.sample (reference
// slimdqn/sample_collection/replay_buffer.py:223-229): the reference keeps snappy-compressed
// elements in a host OrderedDict; here every element's (state, next_state) pair sits uncompressed in
// HBM at slot = key % capacity, so the "decompress + stack" is a pure row gather.  HBM-bound byte
// copy: 2 * n * obs_bytes read + written, 16 B per lane, one workgroup per (sample, half).
#include "common.h"

__global__ __launch_bounds__(256) void k_replay_gather(const uint8_t* __restrict__ store, long obs_bytes,
                                                       const int32_t* __restrict__ slots, uint8_t* __restrict__ s_out,
                                                       uint8_t* __restrict__ n_out) {
    const int b = blockIdx.x, half = blockIdx.y;
    const uint8_t* src = store + ((long)slots[b] * 2 + half) * obs_bytes;
    uint8_t* dst = (half ? n_out : s_out) + (long)b * obs_bytes;
    const long nvec = obs_bytes >> 4;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (long i = threadIdx.x; i < nvec; i += 256) d4[i] = s4[i];
        for (long i = (nvec << 4) + threadIdx.x; i < obs_bytes; i += 256) dst[i] = src[i];
    } else {
        for (long i = threadIdx.x; i < obs_bytes; i += 256) dst[i] = src[i];
    }
}

__global__ void k_replay_gather_scalars(const int32_t* __restrict__ a_store, const float* __restrict__ r_store,
                                        const uint8_t* __restrict__ t_store, const int32_t* __restrict__ slots, int n,
                                        int32_t* __restrict__ a_out, float* __restrict__ r_out,
                                        uint8_t* __restrict__ t_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = slots[i];
    a_out[i] = a_store[s];
    r_out[i] = r_store[s];
    t_out[i] = t_store[s];
}

extern "C" int replay_gather(const uint8_t* store_dev, int64_t obs_bytes, const int32_t* slots_dev, int32_t n,
                             uint8_t* state_out_dev, uint8_t* next_state_out_dev, void* stream) {
    IDQN_REQUIRE(store_dev && slots_dev && state_out_dev && next_state_out_dev, "replay_gather: null pointer");
    IDQN_REQUIRE(n >= 1 && obs_bytes >= 1, "replay_gather: n = %d, obs_bytes = %ld", n, (long)obs_bytes);
    hipLaunchKernelGGL(k_replay_gather, dim3(n, 2), dim3(256), 0, (hipStream_t)stream, store_dev, (long)obs_bytes,
                       slots_dev, state_out_dev, next_state_out_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int replay_gather_scalars(const int32_t* action_store_dev, const float* reward_store_dev,
                                     const uint8_t* terminal_store_dev, const int32_t* slots_dev, int32_t n,
                                     int32_t* action_out_dev, float* reward_out_dev, uint8_t* terminal_out_dev,
                                     void* stream) {
    IDQN_REQUIRE(action_store_dev && reward_store_dev && terminal_store_dev && slots_dev && n >= 1,
                 "replay_gather_scalars: bad arguments");
    hipLaunchKernelGGL(k_replay_gather_scalars, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       action_store_dev, reward_store_dev, terminal_store_dev, slots_dev, n, action_out_dev,
                       reward_out_dev, terminal_out_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}
