// Plane-layout convolution, forward and data gradient (convp.h has the layouts and the arithmetic).
//   forward        architectures/dqn.py:42-52  flax nn.Conv (NHWC x HWIO, cross-correlation) + bias + ReLU
//   data gradient  jax.value_and_grad, idqn.py:105: a stride-1 convolution over the zero-bordered dout planes with the
//                  re-indexed kernel (packed by k_stage), one variant per output parity; ReLU mask of the layer below
//
// Work decomposition: a workgroup (4 waves, one per SIMD, ONE workgroup per CU) owns `np` consecutive output positions of
// one (net, batch block) for all CO channels: np * CT tiles of 32 samples x 32 channels, dealt round-robin to the waves
// (<= NT each).  The host sizes the items so that one launch is (about) one workgroup per CU with equal work.
// K loop: a superstep = one kernel row kh and one 16-channel chunk.  Per superstep the workgroup stages by LDS-DMA
//   * the NQ taps x CT tiles x 3 planes of packed weights (one contiguous run), and
//   * for every input row its positions touch, the STRIP of pixel chunks they read -- neighbouring output positions
//     share pixels (3x3 stride 1: each staged chunk serves 3 taps), which is what keeps the L2 -> LDS traffic per MFMA
//     low enough (~25 B/clk/CU) for the matrix cores to be the limit;
// double-buffered, one s_barrier per superstep, then NQ x NT tile-steps of 6 (Conv_0: 3) MFMAs from LDS fragments.
// Orientation: A = activations (rows = samples), B = weights (columns = channels), so a lane ends up with 4 x 4
// consecutive samples of ONE channel: plane rows are written as 8-byte pieces, the bias / mask is one value per lane.
#include "convp.h"

namespace {

__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* p) {
    auto q = (__attribute__((address_space(3))) s16x4*)p;
    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q);
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q + 32);  // 4 rows (256 B) on
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, v0), __builtin_bit_cast(bf16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 frag_lin(const unsigned char* p) {
    return *(const __attribute__((address_space(3))) bf16x8*)p;
}

template <int NPA, int CT, int NQ, int NT>
__global__ __launch_bounds__(256) void k_cfwd(CFwdArgs a, unsigned stage_bytes) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int WB = NQ * CT * 3 * 1024, BLKA = NPA * 1024, NWP = NQ * CT * 3;
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, cl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const CItem it = a.items[xcd_contiguous_id()];  // an XCD walks consecutive items of one net
    const CVar& v = a.var[it.var];
    const int OW = v.OW, p0 = it.p0, np = it.np;
    const int in_slot = (a.in_split > 0 ? (it.net >= a.in_split ? 1 : 0) : it.net) * a.nb + it.bb;
    const int out_slot = it.net * a.nb + it.bb;
    const unsigned long in_base = (unsigned long)a.in + (unsigned long)in_slot * a.in_slot;
    const int oh0 = p0 / OW;
    // strips: input row r of this workgroup serves output columns [c0, c0 + ncol) of output row oh0 + r
    int nx[CP_MAX_STRIPS], c0[CP_MAX_STRIPS];
    unsigned soff[CP_MAX_STRIPS];
    unsigned long sb[CP_MAX_STRIPS];
    {
        unsigned accb = 0;
#pragma unroll
        for (int r = 0; r < CP_MAX_STRIPS; ++r) {
            const int row = oh0 + r;
            const int lo = max(p0, row * OW), hi = min(p0 + np, (row + 1) * OW);
            const int ncol = hi - lo;
            c0[r] = lo - row * OW;
            nx[r] = ncol > 0 ? (ncol - 1) * a.SX + NQ : 0;
            soff[r] = accb;
            accb += (unsigned)nx[r] * BLKA;
            sb[r] = in_base + (unsigned long)(row * a.S + v.in_off_h) * (unsigned long)a.row_bytes +
                    (unsigned long)(c0[r] * a.S + v.in_off_w) * (unsigned long)a.pix_bytes;
        }
    }
    // tiles of this wave: tile index wave + 4 i -> position j = idx / CT, channel tile ct = idx % CT (= wave % CT)
    const int ct = wave % CT;
    unsigned abase[NT];
    int tpos[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int j = (wave + 4 * i) / CT;
        tpos[i] = j < np ? p0 + j : -1;
        const int p = p0 + min(j, np - 1);
        const int oh = p / OW, ow = p - oh * OW, r = oh - oh0;
        int c0r = c0[0];
        unsigned so = soff[0];
#pragma unroll
        for (int s = 1; s < CP_MAX_STRIPS; ++s) {
            c0r = (r == s) ? c0[s] : c0r;
            so = (r == s) ? soff[s] : so;
        }
        abase[i] = WB + so + (unsigned)((ow - c0r) * a.SX) * BLKA;
    }
    const unsigned lo_tr = (8 * h + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&lds[0];
    const unsigned long wb0 = (unsigned long)a.wq + (unsigned long)it.net * a.wq_stride + (unsigned long)v.w_off;
    const int NSS = a.KH * a.NCC;

    auto stage = [&](int ss, unsigned buf) {
        const int kh = ss / a.NCC, cc = ss - kh * a.NCC;
        const unsigned long wsrc = wb0 + (unsigned long)ss * WB;
        for (int i = wave; i < NWP; i += 4) dma16(lane16, wsrc + (unsigned long)i * 1024, buf + i * 1024);
        const unsigned long so_ = (unsigned long)kh * (unsigned long)a.row_bytes + (unsigned long)cc * 1024;
#pragma unroll
        for (int r = 0; r < CP_MAX_STRIPS; ++r) {
            const unsigned long src = sb[r] + so_;
            const unsigned dst = buf + WB + soff[r];
            for (int x = wave; x < nx[r]; x += 4) {
#pragma unroll
                for (int pl = 0; pl < NPA; ++pl)
                    dma16(lane16, src + (unsigned long)x * (unsigned long)a.xstep + (unsigned long)pl * (unsigned long)a.plane_bytes,
                          dst + (x * NPA + pl) * 1024);
            }
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    stage(0, lds0);
    for (int ss = 0; ss < NSS; ++ss) {
        const unsigned cur_off = (ss & 1) * stage_bytes;
        // this wave's copies of superstep ss have landed; the barrier tells it so has everybody else's, and that
        // nobody still reads the other buffer (superstep ss - 1), which is re-filled next
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ss + 1 < NSS) stage(ss + 1, lds0 + ((ss + 1) & 1) * stage_bytes);
        const unsigned char* cur = lds + cur_off;
        bf16x8 wf[2][3], af[2][NPA];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[0][pl] = frag_lin(cur + (ct * 3 + pl) * 1024 + lane16);
#pragma unroll
        for (int pl = 0; pl < NPA; ++pl) af[0][pl] = frag_tr(cur + abase[0] + pl * 1024 + lo_tr);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int u = q * NT + i;
                // fragments of the next tile-step are requested before this one's MFMAs
                if (i == NT - 1 && q + 1 < NQ) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        wf[(q + 1) & 1][pl] = frag_lin(cur + (((q + 1) * CT + ct) * 3 + pl) * 1024 + lane16);
                }
                if (u + 1 < NQ * NT) {
                    const int qn = (u + 1) / NT, in_ = (u + 1) % NT;
#pragma unroll
                    for (int pl = 0; pl < NPA; ++pl)
                        af[(u + 1) & 1][pl] = frag_tr(cur + abase[in_] + qn * BLKA + pl * 1024 + lo_tr);
                }
                const bf16x8* A = af[u & 1];
                const bf16x8* W = wf[q & 1];
                if (NPA == 3) {  // smallest terms first
                    acc[i] = mfma_bf16(A[2], W[0], acc[i]);
                    acc[i] = mfma_bf16(A[0], W[2], acc[i]);
                    acc[i] = mfma_bf16(A[1], W[1], acc[i]);
                    acc[i] = mfma_bf16(A[1], W[0], acc[i]);
                    acc[i] = mfma_bf16(A[0], W[1], acc[i]);
                    acc[i] = mfma_bf16(A[0], W[0], acc[i]);
                } else {
                    acc[i] = mfma_bf16(A[0], W[2], acc[i]);
                    acc[i] = mfma_bf16(A[0], W[1], acc[i]);
                    acc[i] = mfma_bf16(A[0], W[0], acc[i]);
                }
            }
        }
    }

    // ---- epilogue: lane = channel co, registers = samples (r & 3) + 8 (r >> 2) + 4 h -------------------------------
    const int co = ct * 32 + cl;
    float bias = 0.f;
    if (a.epilogue == 0) bias = a.wbase[it.net][a.b_off + co];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (tpos[i] < 0) continue;  // wave-uniform
        const int p = tpos[i];
        const int oh = p / OW, ow = p - oh * OW;
        const int yh = oh * v.out_mul + v.out_add_h, yw = ow * v.out_mul + v.out_add_w;
        float val[16];
        if (a.epilogue == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) val[r] = fmaxf(acc[i][r] + bias, 0.f);
        } else {
            const unsigned short* M = a.mask3 + ((unsigned long)out_slot * a.mask_slot) / 2 +
                                      ((long)(yh + a.mask_lo_h) * a.mask_Wp + (yw + a.mask_lo_w)) * (3L * a.mask_C * 32) +
                                      (long)co * 32 + 4 * h;
            uint2 mk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) mk[g] = *reinterpret_cast<const uint2*>(M + 8 * g);
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned w0 = mk[g].x, w1 = mk[g].y;
                const float m0 = __uint_as_float(w0 << 16), m1 = __uint_as_float(w0 & 0xffff0000u);
                const float m2 = __uint_as_float(w1 << 16), m3 = __uint_as_float(w1 & 0xffff0000u);
                val[4 * g + 0] = m0 > 0.f ? acc[i][4 * g + 0] : 0.f;
                val[4 * g + 1] = m1 > 0.f ? acc[i][4 * g + 1] : 0.f;
                val[4 * g + 2] = m2 > 0.f ? acc[i][4 * g + 2] : 0.f;
                val[4 * g + 3] = m3 > 0.f ? acc[i][4 * g + 3] : 0.f;
            }
            if (a.pb) {  // sum over the 32 samples, fixed order: registers, then the two half-waves
#pragma unroll
                for (int r = 0; r < 16; ++r) s += val[r];
                s += __shfl_xor(s, 32);
                if (h == 0) a.pb[((long)out_slot * (a.out_H * a.out_W) + (long)yh * a.out_W + yw) * a.CO + co] = s;
            }
        }
        if (a.out3) {
            unsigned short* O = a.out3 + ((unsigned long)out_slot * a.out_slot) / 2 +
                                ((long)(yh + a.out_lo_h) * a.out_Wp + (yw + a.out_lo_w)) * (3L * a.CO * 32) + (long)co * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned q0a, q1a, q2a, q0b, q1b, q2b;
                split3_pk(val[4 * g + 0], val[4 * g + 1], q0a, q1a, q2a);
                split3_pk(val[4 * g + 2], val[4 * g + 3], q0b, q1b, q2b);
                *reinterpret_cast<uint2*>(O + 8 * g) = make_uint2(q0a, q0b);
                *reinterpret_cast<uint2*>(O + (long)a.CO * 32 + 8 * g) = make_uint2(q1a, q1b);
                *reinterpret_cast<uint2*>(O + 2L * a.CO * 32 + 8 * g) = make_uint2(q2a, q2b);
            }
        }
        if (a.out_f32) {
            float* F = a.out_f32 + (long)out_slot * a.f32_slot + (((long)yh * a.f32_W + yw) * a.CO + co) * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(F + 8 * g) = make_float4(val[4 * g], val[4 * g + 1], val[4 * g + 2], val[4 * g + 3]);
        }
    }
}

template <int NPA, int CT, int NQ, int NT>
int launch_one(const CFwdArgs& a, int n_items, size_t lds_bytes, hipStream_t q) {
    static size_t attr = 0;  // per instantiation
    if (lds_bytes > attr) {
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_cfwd<NPA, CT, NQ, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr = lds_bytes;
    }
    hipLaunchKernelGGL((k_cfwd<NPA, CT, NQ, NT>), dim3((unsigned)n_items), dim3(256), lds_bytes, q, a, (unsigned)(lds_bytes / 2));
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

template <int NPA, int CT, int NQ>
int launch_nt(const CFwdArgs& a, int NT, int n_items, size_t lds_bytes, hipStream_t q) {
    switch (NT) {
        case 1: return launch_one<NPA, CT, NQ, 1>(a, n_items, lds_bytes, q);
        case 2: return launch_one<NPA, CT, NQ, 2>(a, n_items, lds_bytes, q);
        case 3: return launch_one<NPA, CT, NQ, 3>(a, n_items, lds_bytes, q);
        default: break;
    }
    if (CT == 1) {
        switch (NT) {
            case 4: return launch_one<NPA, 1, NQ, 4>(a, n_items, lds_bytes, q);
            case 5: return launch_one<NPA, 1, NQ, 5>(a, n_items, lds_bytes, q);
            case 6: return launch_one<NPA, 1, NQ, 6>(a, n_items, lds_bytes, q);
            default: break;
        }
    }
    IDQN_REQUIRE(false, "plane conv: %d tiles per wave with %d channel tiles is not built", NT, CT);
}

}  // namespace

int convp_fwd_max_nt(int CT) { return CT == 1 ? 6 : 3; }

int convp_launch_fwd(const CFwdArgs& a, int NPA, int CT, int NQ, int NT, int n_items, size_t lds_bytes, hipStream_t q) {
    IDQN_REQUIRE(lds_bytes <= 160 * 1024, "plane conv: %zu bytes of LDS per workgroup", lds_bytes);
    if (NPA == 1 && NQ == 2) return CT == 1 ? launch_nt<1, 1, 2>(a, NT, n_items, lds_bytes, q) : launch_nt<1, 2, 2>(a, NT, n_items, lds_bytes, q);
    if (NPA == 3 && NQ == 2) return CT == 1 ? launch_nt<3, 1, 2>(a, NT, n_items, lds_bytes, q) : launch_nt<3, 2, 2>(a, NT, n_items, lds_bytes, q);
    if (NPA == 3 && NQ == 3) return CT == 1 ? launch_nt<3, 1, 3>(a, NT, n_items, lds_bytes, q) : launch_nt<3, 2, 3>(a, NT, n_items, lds_bytes, q);
    if (NPA == 3 && NQ == 4) return CT == 1 ? launch_nt<3, 1, 4>(a, NT, n_items, lds_bytes, q) : launch_nt<3, 2, 4>(a, NT, n_items, lds_bytes, q);
    IDQN_REQUIRE(false, "plane conv: no kernel for %d planes, %d taps per superstep", NPA, NQ);
}
