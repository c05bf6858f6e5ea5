// HBM replay store: gather of sampled slots into a contiguous minibatch.
//
// Replaces the fetch + unpack + np.stack of ReplayBuffer.sample (reference
// slimdqn/sample_collection/replay_buffer.py:223-229): the reference keeps snappy-compressed
// elements in a host OrderedDict; here every element's (state, next_state) pair sits uncompressed in
// HBM at slot = key % capacity, so the "decompress + stack" is a pure row gather.  HBM-bound byte
// copy: 2 * n * obs_bytes read + written, 16 B per lane, one workgroup per (sample, half).
#include "common.h"
#include <algorithm>

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

// ---------------------------------------------------------------------------------------------------
// Frame-ring store: every environment frame is written to HBM ONCE (ring slot = transition index % n_frames);
// a replay element is 8 int32 of metadata and its frame stacks are assembled at sample time -- the device-side
// counterpart of the reference's stack building (replay_buffer.py:119-137: `state[..., ch] = observation`, zero
// frames before the episode start) fused with the gather + np.stack of sample() (:223-229).
// meta row: {newest state frame slot, valid state frames, newest next_state frame slot, valid next frames,
//            action, reward (f32 bits), terminal, 0}.
// Output element (pixel, ch) of a stack = frame[newest - (stack-1-ch)][pixel], or 0 where ch < stack - valid.
// grid (sample, state|next, chunk); the uint8 x 4-stack case packs one u32 per pixel (16 B per lane).
__global__ __launch_bounds__(256) void k_replay_gather_stacked(const uint8_t* __restrict__ frames, long n_frames,
                                                               long frame_elems, int itemsize, int stack,
                                                               const int32_t* __restrict__ meta,
                                                               const int32_t* __restrict__ slots,
                                                               uint8_t* __restrict__ s_out, uint8_t* __restrict__ n_out,
                                                               int32_t* __restrict__ a_out, float* __restrict__ r_out,
                                                               uint8_t* __restrict__ t_out) {
    const int b = blockIdx.x, half = blockIdx.y;
    const int32_t* m = meta + (long)slots[b] * 8;
    const long newest = m[2 * half], valid = m[2 * half + 1];
    const long frame_bytes = frame_elems * itemsize;
    uint8_t* dst = (half ? n_out : s_out) + (long)b * frame_bytes * stack;
    if (half == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        a_out[b] = m[4];
        r_out[b] = __int_as_float(m[5]);
        t_out[b] = (uint8_t)m[6];
    }
    const long tid = (long)blockIdx.z * 256 + threadIdx.x, nthr = (long)gridDim.z * 256;
    if (itemsize == 1 && stack == 4 && (frame_bytes & 15) == 0 && ((uintptr_t)frames & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
        const uint4* f[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const long back = 3 - ch;
            f[ch] = back < valid ? reinterpret_cast<const uint4*>(frames + ((newest - back + n_frames) % n_frames) * frame_bytes)
                                 : nullptr;
        }
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (long i = tid; i < (frame_bytes >> 4); i += nthr) {  // 16 pixels per lane: 4 x 16 B in, 64 B out
            uint4 v[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) v[ch] = f[ch] ? f[ch][i] : zero;
            const uint32_t w[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                                      {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // 4 pixels x 4 channels: byte transpose of a 4 x 4 block
                const uint32_t a = w[g][0], b = w[g][1], c = w[g][2], d = w[g][3];
                uint4 o;
                o.x = (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | (d << 24);
                o.y = ((a >> 8) & 0xff) | (b & 0xff00) | ((c & 0xff00) << 8) | ((d & 0xff00) << 16);
                o.z = ((a >> 16) & 0xff) | ((b >> 8) & 0xff00) | (c & 0xff0000) | ((d & 0xff0000) << 8);
                o.w = (a >> 24) | ((b >> 16) & 0xff00) | ((c >> 8) & 0xff0000) | (d & 0xff000000);
                d4[4 * i + g] = o;
            }
        }
        return;
    }
    if (itemsize == 1 && stack == 4 && (frame_bytes & 3) == 0 && ((uintptr_t)frames & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
        const uint32_t* f[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const long back = 3 - ch;
            f[ch] = back < valid ? reinterpret_cast<const uint32_t*>(frames + ((newest - back + n_frames) % n_frames) * frame_bytes)
                                 : nullptr;
        }
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (long i = tid; i < (frame_bytes >> 2); i += nthr) {  // 4 pixels per lane
            const uint32_t a = f[0] ? f[0][i] : 0u, b = f[1] ? f[1][i] : 0u, c = f[2] ? f[2][i] : 0u, d = f[3] ? f[3][i] : 0u;
            uint4 o;
            o.x = (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | (d << 24);
            o.y = ((a >> 8) & 0xff) | (b & 0xff00) | ((c & 0xff00) << 8) | ((d & 0xff00) << 16);
            o.z = ((a >> 16) & 0xff) | ((b >> 8) & 0xff00) | (c & 0xff0000) | ((d & 0xff0000) << 8);
            o.w = (a >> 24) | ((b >> 16) & 0xff00) | ((c >> 8) & 0xff0000) | (d & 0xff000000);
            d4[i] = o;
        }
        return;
    }
    const long total = frame_bytes * stack;  // generic: any element size / stack depth, one byte per iteration
    for (long o = tid; o < total; o += nthr) {
        const long e = o / itemsize, byte = o - e * itemsize;
        const long pix = e / stack, ch = e - pix * stack, back = stack - 1 - ch;
        uint8_t v = 0;
        if (back < valid) v = frames[((newest - back + n_frames) % n_frames) * frame_bytes + pix * itemsize + byte];
        dst[o] = v;
    }
}

extern "C" int replay_gather_stacked(const uint8_t* frames_dev, int64_t n_frames, int64_t frame_elems, int32_t itemsize,
                                     int32_t stack, const int32_t* meta_dev, const int32_t* slots_dev, int32_t n,
                                     uint8_t* state_out_dev, uint8_t* next_state_out_dev, int32_t* action_out_dev,
                                     float* reward_out_dev, uint8_t* terminal_out_dev, void* stream) {
    IDQN_REQUIRE(frames_dev && meta_dev && slots_dev && state_out_dev && next_state_out_dev && action_out_dev &&
                     reward_out_dev && terminal_out_dev, "replay_gather_stacked: null pointer");
    IDQN_REQUIRE(n >= 1 && n_frames >= 1 && frame_elems >= 1 && itemsize >= 1 && itemsize <= 16 && stack >= 1 && stack <= 64,
                 "replay_gather_stacked: n = %d, n_frames = %ld, frame_elems = %ld, itemsize = %d, stack = %d", n,
                 (long)n_frames, (long)frame_elems, itemsize, stack);
    const long bytes = (long)frame_elems * itemsize * stack;
    // one iteration per lane where possible: a lane emits 64 B (16 pixels x 4 channels) on the packed path
    const int chunks = (int)std::min<long>(16, std::max<long>(1, (bytes + 64 * 256 - 1) / (64 * 256)));
    hipLaunchKernelGGL(k_replay_gather_stacked, dim3(n, 2, chunks), dim3(256), 0, (hipStream_t)stream, frames_dev,
                       (long)n_frames, (long)frame_elems, itemsize, stack, meta_dev, slots_dev, state_out_dev,
                       next_state_out_dev, action_out_dev, reward_out_dev, terminal_out_dev);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
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

// ReplayBuffer.add's device half (slimdqn/sample_collection/replay_buffer.py:206-213: the reference stores the
// transition on the host): the newest frame goes from a PINNED host staging slot to its slot of the frame ring in HBM,
// asynchronously on the caller's stream.  The caller owns the reuse discipline of the staging slots.
extern "C" int replay_add_frame(void* frame_ring_dev, int64_t slot, int64_t frame_bytes, const void* frame_host_pinned,
                                void* stream) {
    IDQN_REQUIRE(frame_ring_dev && frame_host_pinned && slot >= 0 && frame_bytes > 0, "replay_add_frame: bad arguments");
    IDQN_HIP_CHECK(hipMemcpyAsync((char*)frame_ring_dev + slot * frame_bytes, frame_host_pinned, (size_t)frame_bytes,
                                  hipMemcpyHostToDevice, (hipStream_t)stream));
    return IDQN_OK;
}

// Growing the frame ring (rare: element-less transitions used up the slack): the live frames, transition indices
// [first_t, first_t + count), move from slot t % old_n of the old ring to slot t % new_n of the new one.
__global__ __launch_bounds__(256) void k_ring_regrow(const uint8_t* __restrict__ old_ring, long old_n,
                                                     uint8_t* __restrict__ new_ring, long new_n, long first_t,
                                                     long frame_bytes) {
    const long t = first_t + blockIdx.x;
    const uint8_t* src = old_ring + (t % old_n) * frame_bytes;
    uint8_t* dst = new_ring + (t % new_n) * frame_bytes;
    if ((frame_bytes & 15) == 0 && (((uintptr_t)old_ring | (uintptr_t)new_ring) & 15) == 0) {
        for (long i = threadIdx.x; i < frame_bytes / 16; i += 256)
            reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    } else {
        for (long i = threadIdx.x; i < frame_bytes; i += 256) dst[i] = src[i];
    }
}

extern "C" int replay_ring_regrow(const void* old_ring_dev, int64_t old_n, void* new_ring_dev, int64_t new_n,
                                  int64_t first_t, int64_t count, int64_t frame_bytes, void* stream) {
    IDQN_REQUIRE(old_ring_dev && new_ring_dev && old_ring_dev != new_ring_dev, "replay_ring_regrow: two distinct rings required");
    IDQN_REQUIRE(old_n >= 1 && new_n >= old_n && first_t >= 0 && count >= 0 && count <= old_n && frame_bytes >= 1,
                 "replay_ring_regrow: old_n = %ld, new_n = %ld, first_t = %ld, count = %ld, frame_bytes = %ld", (long)old_n,
                 (long)new_n, (long)first_t, (long)count, (long)frame_bytes);
    for (int64_t done = 0; done < count; done += 1 << 30) {  // grid.x limit; one launch in practice
        const int64_t n = std::min<int64_t>(count - done, 1 << 30);
        hipLaunchKernelGGL(k_ring_regrow, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)old_ring_dev,
                           (long)old_n, (uint8_t*)new_ring_dev, (long)new_n, (long)(first_t + done), (long)frame_bytes);
        IDQN_HIP_CHECK(hipGetLastError());
    }
    return IDQN_OK;
}
