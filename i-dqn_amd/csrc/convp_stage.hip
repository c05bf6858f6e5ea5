// Staging launch of the plane-layout conv path: one grid does both things the first conv needs and nothing else does.
//   blocks [0, n_prep_blocks)   uint8 NHWC minibatch (state, next_state) -> one bf16 plane, batch-minor, zero-bordered
//                               (the (s, s') minibatch tile is transposed through LDS: coalesced on both sides).
//                               architectures/dqn.py:44 divides by 255; here the raw pixel value (exact in bf16) is
//                               stored and the Conv_0 kernel is packed as w / 255 instead.
//   the rest                    every conv kernel of every net (HWIO f32, idqn.py:48-50 leaves) -> three bf16 planes in
//                               MFMA-fragment order (convp.h), forward kernels of the 2K nets and the re-indexed
//                               data-gradient kernels of the K online nets (per output parity for the stride-2 Conv_1):
//                               kh = (r + PL) % S + S * (K/S - 1 - kh') for output parity r (a plain flip when S == 1).
#include "convp.h"

namespace {

struct NoSlots { int32_t slot[1]; };
template <bool REPLAY, typename SL>
__global__ __launch_bounds__(256) void k_stage(StageArgs a, SL sl) {
    // [element][sample], 80-byte rows keep the 16-byte reads aligned.  The four 8-sample groups of a row are XOR-swizzled
    // by (row >> 3) & 3: a wave's transposing 2-byte stores go to 8 rows that are 8 apart (640 bytes = 0 mod 32 banks) --
    // 8-way conflicts unswizzled (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.81 in round 3), 2-way with the groups spread
    __shared__ unsigned short tile[64][40];
    const int t = threadIdx.x;
    warm_kernargs<sizeof(StageArgs)>();
    if ((int)blockIdx.x < a.n_prep_blocks) {
        const int n_et = (int)((a.E + 63) / 64);
        int b = blockIdx.x;
        const int et = b % n_et;
        b /= n_et;
        const int bb = b % a.nb, set = b / a.nb;
        const uint8_t* src = a.src[set];
        const long e0 = (long)et * 64;
        if (REPLAY) {
            // the 64 elements of this tile = 16 pixels x 4 stacked frames: thread (sample s, channel c8 < 4) reads the 16
            // pixels of ITS frame as one 16-byte load (replay_buffer.py:223-229 stacks along the last axis)
            const int s = t >> 3, c8 = t & 7, bg = bb * 32 + s;
            if (c8 < 4) {
                uint4 w = make_uint4(0u, 0u, 0u, 0u);
                if (bg < a.B) {
                    const int32_t* m = a.rows + (long)sl.slot[bg] * 8;
                    const long newest = m[2 * set], valid = m[2 * set + 1], back = 3 - c8;
                    if (back < valid)
                        w = *reinterpret_cast<const uint4*>(a.frames + ((newest - back + a.n_frames) % a.n_frames) * a.frame_bytes + (long)et * 16);
                    if (et == 0 && set == 0 && c8 == 0) {
                        a.act_out[bg] = m[4];
                        a.rew_out[bg] = __int_as_float(m[5]);
                        a.term_out[bg] = (uint8_t)m[6];
                    }
                }
                const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int i = 0; i < 16; ++i) {  // pixel i, channel c8: tile row 4 i + c8, (row >> 3) & 3 = (i >> 1) & 3
                    const unsigned u = (ww[i >> 2] >> (8 * (i & 3))) & 0xffu;
                    tile[4 * i + c8][(((s >> 3) ^ ((i >> 1) & 3)) << 3) | (s & 7)] = (unsigned short)(__float_as_uint((float)u) >> 16);  // exact
                }
            }
        } else {
            const int s = t >> 3, c8 = t & 7, bg = bb * 32 + s;
            const long e8 = e0 + c8 * 8;
            unsigned u[8];
            if (bg < a.B && e8 + 8 <= a.E && (((uintptr_t)src + (long)bg * a.E + e8) & 7) == 0) {
                const uint2 w = *reinterpret_cast<const uint2*>(src + (long)bg * a.E + e8);
#pragma unroll
                for (int i = 0; i < 8; ++i) u[i] = ((i < 4 ? w.x : w.y) >> (8 * (i & 3))) & 0xffu;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) u[i] = (bg < a.B && e8 + i < a.E) ? src[(long)bg * a.E + e8 + i] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)  // row c8 * 8 + i: (row >> 3) & 3 = c8 & 3
                tile[c8 * 8 + i][(((s >> 3) ^ (c8 & 3)) << 3) | (s & 7)] = (unsigned short)(__float_as_uint((float)u[i]) >> 16);  // exact
        }
        __syncthreads();
        const int row = t >> 2, part = t & 3;
        const long e = e0 + row;
        if (e < a.E) {
            const int WC = a.W * a.C, E32 = (int)e;
            const int hh = E32 / WC, r = E32 - hh * WC, w = r / a.C, c = r - w * a.C;
            const long orow = ((long)(hh + a.lo_h) * a.Wp + (w + a.lo_w)) * a.C + c;
            unsigned short* dst = a.x1 + (((long)set * a.nb + bb) * a.Hp * a.Wp * a.C + orow) * 32 + part * 8;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&tile[row][(part ^ ((row >> 3) & 3)) * 8]);
        }
        return;
    }
    const long pb = (long)blockIdx.x - a.n_prep_blocks;
    if (pb == 0 && a.bcinv) {
        for (int k = t; k < a.K; k += 256) {  // (any number of heads)
            const double tt = (double)(a.count[k] + 1);
            a.bcinv[2 * k] = 1.0f / (1.0f - (float)pow((double)a.b1, tt));
            a.bcinv[2 * k + 1] = 1.0f / (1.0f - (float)pow((double)a.b2, tt));
        }
    }
    int ji = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < a.n_jobs && pb >= a.job[i].first_block) ji = i;
    const PackJob& j = a.job[ji];
    const int local = (int)(pb - j.first_block);
    const int net = local / j.blocks_per_net, blk = local - net * j.blocks_per_net;
    if (net >= j.n_nets) return;
    int e = blk * 256 + t;
    if (e >= j.KHv * j.NCC * j.NQ * j.CT * 64) return;
    const int lane = e & 63;
    e >>= 6;
    const int ct = e % j.CT;
    e /= j.CT;
    const int q = e % j.NQ, ss = e / j.NQ;
    const int khv = ss / j.NCC, cc = ss - khv * j.NCC;
    const int hh = lane >> 5, col = ct * 32 + (lane & 31);
    const float* W = a.wbase[net] + j.src_off;
    float v[8];
    if (j.mode == 0) {  // k runs over kernel rows (stride CO); lanes over out channels: 8 coalesced loads
        const int rho0 = (j.CI >= 16 ? q * j.CI + cc * 16 : q * 16) + 8 * hh;
        const float* src = W + ((long)khv * j.KW * j.CI + rho0) * j.CO + col;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = src[(long)i * j.CO];
    } else {  // re-indexed kernel: k runs over the layer's OUT channels, contiguous in HWIO: two 16-byte loads
        const int kh = (j.rh + j.PLh) % j.S + j.S * (j.KHs - 1 - khv);
        const int kw = (j.rw + j.PLw) % j.S + j.S * (j.KHs - 1 - q);
        const float* src = W + ((long)(kh * j.KW + kw) * j.CI + col) * j.CO + cc * 16 + 8 * hh;
        const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    }
    if (j.div255) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] / 255.0f;
    }
    unsigned p0[4], p1[4], p2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split3_pk(v[2 * i], v[2 * i + 1], p0[i], p1[i], p2[i]);
    unsigned char* dst = (unsigned char*)a.wq + (long)net * a.wq_stride + j.dst_off +
                         (((long)(ss * j.NQ + q) * j.CT + ct) * 3) * 1024 + lane * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
    *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
}

}  // namespace

int convp_launch_stage(const StageArgs& a, int n_blocks, hipStream_t q, const StageSlots* slots) {
    if (a.frames && slots) hipLaunchKernelGGL((k_stage<true, StageSlots>), dim3((unsigned)n_blocks), dim3(256), 0, q, a, *slots);
    else hipLaunchKernelGGL((k_stage<false, NoSlots>), dim3((unsigned)n_blocks), dim3(256), 0, q, a, NoSlots{{0}});
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}
