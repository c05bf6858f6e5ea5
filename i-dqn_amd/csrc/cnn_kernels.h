// gfx950 kernels of the Nature-CNN i-DQN gradient step (included by qnet.hip).
//
// Data layout in HBM (one choice drives every kernel): activations are BATCH-MINOR,
//     act[net][batch_block][row = (h, w, c)][32 samples]            (f32)
// i.e. a "row" is one (pixel, channel) for a block of 32 samples = 128 B.  With the batch on the
// 32-wide side of v_mfma_f32_32x32x2_f32, the operands of the forward / data-gradient contractions are
// plain 128-byte rows (weights [k][out] row-major, activations [k][32]) and results store as whole rows:
//   forward      D[i = out channel][j = sample] += W[k][i] * act[k][j]          (rows of W, rows of act)
//   data grad    the same kernel on the zero-bordered dout buffer with flipped / transposed weights
//   weight grad  D[i = in  row   ][j = out row] += act[i][b] * dout[j][b]       (k = sample: each lane needs
//                16 floats of ITS OWN row -> rows go through an LDS image padded to 36 floats per row)
// Spatial SAME padding (flax default, architectures/dqn.py:43-51) is materialised as zero borders of
// the activation buffers, so no kernel has a bounds branch in its k-loop.
//
// 256-thread workgroups (4 waves).  Conv forward / data gradient: LDS-DMA staged k-chunks shared by the 4
// waves.  Dense_0: weight-streaming kernels (forward: registers, double-buffered; data gradient: dh staged in
// LDS; weight gradient + Adam: MFMA tile parked in LDS, then whole-row streaming of theta / m / v).
// f32 MFMA is 64 FLOP/clk/SIMD: one dword of each operand per 64-cycle instruction.
#pragma once
#include <type_traits>
#include "convp.h"

#include "dense0_update.h"


// --------------------------------------------------------------------------------------------
// input staging: uint8 NHWC minibatch -> f32 / 255 batch-minor, through an LDS tile (the (s, s')
// minibatch tile is transposed in LDS so that both the HBM read and the HBM write are coalesced)
// architectures/dqn.py:44  `jnp.array(x, ndmin=4) / 255.0`
// --------------------------------------------------------------------------------------------
struct PrepArgs {
    const uint8_t* src[2];  // state, next_state  [B][E]
    float* x;               // [n_sets][nb][g.block]
    long E;                 // H*W*C
    int B, nb, n_sets;
    ActGeom g;
};

__global__ __launch_bounds__(256) void k_prep_u8(PrepArgs a) {
    __shared__ float tile[64][33];
    const int t = threadIdx.x;
    const long e0 = (long)blockIdx.x * 64;
    const int bb = blockIdx.y, set = blockIdx.z;
    const uint8_t* src = a.src[set];
    {
        const int b = t >> 3, c = t & 7;
        const int bg = bb * 32 + b;
        const long e8 = e0 + c * 8;
        if (bg < a.B && e8 + 8 <= a.E && (((uintptr_t)src + (long)bg * a.E + e8) & 7) == 0) {
            const uint2 w = *reinterpret_cast<const uint2*>(src + (long)bg * a.E + e8);  // 8 pixels per load
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned u = ((i < 4 ? w.x : w.y) >> (8 * (i & 3))) & 0xffu;
                tile[c * 8 + i][b] = (float)u / 255.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                long e = e8 + i;
                float v = 0.f;
                if (bg < a.B && e < a.E) v = (float)src[(long)bg * a.E + e] / 255.0f;
                tile[c * 8 + i][b] = v;
            }
        }
    }
    __syncthreads();
    float* x = a.x + ((long)set * a.nb + bb) * a.g.block;
    const int b = t & 31;
    // (h, w, c) of this thread's first element by 32-bit division once; its later elements are 8 further on
    const int E32 = (int)a.E, WC = a.g.W * a.g.C;
    int e = (int)e0 + (t >> 5);
    int hh = e / WC, r = e - hh * WC;
    int w = r / a.g.C, c = r - w * a.g.C;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int el = pass * 8 + (t >> 5);
        if (e < E32) {
            const long row = ((long)(hh + a.g.lo_h) * a.g.Wp + (w + a.g.lo_w)) * a.g.C + c;
            x[row * 32 + b] = tile[el][b];
        }
        e += 8;
        c += 8;
        while (c >= a.g.C) { c -= a.g.C; ++w; }
        while (w >= a.g.W) { w -= a.g.W; ++hh; }
    }
}

// --------------------------------------------------------------------------------------------
// conv forward + bias + ReLU  (architectures/dqn.py:42-52; flax nn.Conv: NHWC x HWIO, cross-correlation)
// workgroup = (net, batch block, group of NPW output positions); waves = (out-channel tile, position subset)
// --------------------------------------------------------------------------------------------
// A "variant" is one sub-convolution of a launch.  The forward convs have one; the data gradient of the
// stride-2 Conv_1 is four stride-1 sub-convolutions (one per output parity) folded into one launch.
struct ConvVariant {
    long w_off;                // offset of this variant's [KH][KW*CI][CO] weight block from the net's base
    int in_off_h, in_off_w;    // added to (oh*S, ow*S): first padded input row / column of output (0, 0)
    int OH, OW;                // output extent of the variant
    int out_mul, out_add_h, out_add_w;  // output position (oh, ow) -> (oh*out_mul + out_add_h, ow*out_mul + out_add_w)
    int pg_begin;              // first position group of this variant inside a (net, batch block)
};
struct ConvFwdArgs {
    const float* in;            // [n_in_sets][nb][in_block]  zero-bordered
    float* out;                 // [n_nets][nb][out_block]
    const float* const* wbase;  // [n_nets] parameter base of each net (online or target arena slice) ...
    const float* wt_base;       // ... or, when non-null, a dense [n_nets][wt_stride] buffer of transformed weights
    const int* in_set;          // [n_nets] which input set a net reads
    const float* mask;          // epilogue 1: forward activation whose sign masks the result [n_nets][nb][mask_block]
    long wt_stride, b_off, in_block, out_block, mask_block, n_items;
    int n_nets, nb, npg, n_var, epilogue;  // epilogue 0: + bias, ReLU.   1: * (mask > 0), no bias (data gradient)
    int KH, KWCI, S, CI, CO, IWp;          // KWCI = KW * CI (taps of one kernel row are contiguous rows)
    int out_Wp, out_lo_h, out_lo_w, mask_Wp, mask_lo_h, mask_lo_w;
    ConvVariant var[4];
};

// Workgroup tile = all CO out channels x NPW output positions (x 32 samples) of one (net, batch block).
// Per k-chunk (KC = 32 rows of one kernel row kh: they are CONTIGUOUS both in W[q][co] and in the
// batch-minor activation) the 256 threads copy the weight block (KC x CO) and NPW activation blocks
// (KC x 32) HBM/L2 -> LDS by LDS-DMA, 16 B per lane, double-buffered, one barrier per chunk; the four
// waves then read their MFMA fragments with ds_read_b32 (lanes = consecutive floats: conflict-free).
// Compared with loading fragments straight from L2 this cuts vector-memory instructions per MFMA ~10x
// (the register version was bound by load issue, not by MFMA or bandwidth).
template <int CT, int PPW>  // CT = CO / 32 (1 or 2); PPW = positions per wave
__device__ __forceinline__ void conv_fwd_body(const ConvFwdArgs& a, int item, float* lds) {
    constexpr int NSUB = 4 / CT, NPW = NSUB * PPW, KC = 32, CO = 32 * CT;
    // ONE LDS object (a second one beside an LDS-DMA target makes hipcc wait vmcnt(0) before every
    // ds_read): [buffer][ A: KC x CO | B: NPW x (KC x 32) ]   = 2 * BUF_FL floats
    constexpr int A_FL = KC * CO, B_FL = KC * 32, BUF_FL = A_FL + NPW * B_FL;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, bl = lane & 31, h = lane >> 5;
    int pg = item % a.npg;
    item /= a.npg;
    const int bb = item % a.nb;
    const int n = item / a.nb;
    int vi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < a.n_var && pg >= a.var[i].pg_begin) vi = i;
    const ConvVariant& v = a.var[vi];
    pg -= v.pg_begin;
    const int ct = wave % CT, sub = wave / CT;
    const float* pbase = a.wt_base ? a.wt_base + (long)n * a.wt_stride : a.wbase[n];
    const float* Wg = pbase + v.w_off + t * 4;
    const float* Xg = a.in + ((long)a.in_set[n] * a.nb + bb) * a.in_block + t * 4;
    float* Y = a.out + ((long)n * a.nb + bb) * a.out_block;
    const int npos = v.OH * v.OW;
    long xoff[NPW];
#pragma unroll
    for (int p = 0; p < NPW; ++p) {
        int pos = min(pg * NPW + p, npos - 1);
        int oh = pos / v.OW, ow = pos - oh * v.OW;
        xoff[p] = ((long)(oh * a.S + v.in_off_h) * a.IWp + ow * a.S + v.in_off_w) * a.CI * 32;
    }
    const float bias_on = a.epilogue == 0 ? 1.f : 0.f;
    f32x16 acc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = pbase[a.b_off + ct * 32 + mfma_row(r, h)] * bias_on;  // b_off = 0 when off
    const int JC = a.KWCI / KC, NC = a.KH * JC;
    const long wrow = (long)a.KWCI * CO, xrow = (long)a.IWp * a.CI * 32;
    // LDS-DMA (global_load_lds_dwordx4): every wave copies 1 KiB pieces, lane l <- 16 B at source + 16 l,
    // landing at (wave-uniform LDS base) + 16 l -- the blocks are contiguous, so the LDS image is linear.
    // No VGPR staging, no ds_write.  __syncthreads() drains the DMA (vmcnt(0)) before the barrier.
#define CF_STAGE(c, buf)                                                                      \
    {                                                                                         \
        const int kh_ = (c) / JC, q0_ = ((c) - kh_ * JC) * KC;                                \
        const float* wg_ = Wg + kh_ * wrow + (long)q0_ * CO;                                  \
        const float* xg_ = Xg + kh_ * xrow + (long)q0_ * 32;                                  \
        _Pragma("unroll") for (int i = 0; i < CT; ++i)                                        \
            glds16(wg_ + i * 1024, &lds[(buf) * BUF_FL + i * 1024 + wave * 256]);                          \
        _Pragma("unroll") for (int p = 0; p < NPW; ++p)                                       \
            glds16(xg_ + xoff[p], &lds[(buf) * BUF_FL + A_FL + p * B_FL + wave * 256]);                                   \
    }
    CF_STAGE(0, 0)
    __syncthreads();
    for (int c = 0; c < NC; ++c) {
        const int buf = c & 1;
        if (c + 1 < NC) CF_STAGE(c + 1, buf ^ 1)
        const float* as = &lds[buf * BUF_FL + h * CO + ct * 32 + bl];
        const float* bs = &lds[buf * BUF_FL + A_FL + sub * PPW * B_FL + h * 32 + bl];
#pragma unroll
        for (int k2 = 0; k2 < KC / 2; ++k2) {
            const float av = as[2 * k2 * CO];
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                acc[i] = mfma32(av, bs[i * B_FL + 2 * k2 * 32], acc[i]);
        }
        __syncthreads();
    }
#undef CF_STAGE
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pos = pg * NPW + sub * PPW + i;
        if (pos < npos) {
            const int oh = pos / v.OW, ow = pos - oh * v.OW;
            const int yh = oh * v.out_mul + v.out_add_h, yw = ow * v.out_mul + v.out_add_w;
            const long row0 = ((long)(yh + a.out_lo_h) * a.out_Wp + (yw + a.out_lo_w)) * CO + ct * 32;
            if (a.epilogue == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[(row0 + mfma_row(r, h)) * 32 + bl] = fmaxf(acc[i][r], 0.f);
            } else {
                const float* M = a.mask + ((long)n * a.nb + bb) * a.mask_block +
                                 (((long)(yh + a.mask_lo_h) * a.mask_Wp + (yw + a.mask_lo_w)) * CO + ct * 32) * 32 + bl;
                float mk[16];  // all mask loads before the (may-alias) stores
#pragma unroll
                for (int r = 0; r < 16; ++r) mk[r] = M[mfma_row(r, h) * 32];
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[(row0 + mfma_row(r, h)) * 32 + bl] = mk[r] > 0.f ? acc[i][r] : 0.f;
            }
        }
    }
}
template <int CT, int PPW>
__global__ __launch_bounds__(256) void k_conv_fwd(ConvFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (32 * 32 * CT + (4 / CT) * PPW * 32 * 32)];
    conv_fwd_body<CT, PPW>(a, xcd_contiguous_id(), lds);  // an XCD walks consecutive position groups of one net
}

// Transformed weights for the data gradients-as-forward-convolutions:
//   wt[net][variant][kh'][kw'][co][ci] = W[net][kh(kh', variant)][kw(kw', variant)][ci][co]
// with kh = (r + PL) % S + S * (K/S - 1 - kh') for output parity r (a plain flip when S == 1).  <= 150 KB/head.
struct WtLayer {
    float* wt;  // [K][wt_stride]
    long w_off, wt_stride;
    int KH, KW, CI, CO, S, PLh, PLw, n_var, KHs, KWs;  // KHs x KWs taps per variant
};
struct WtBuildArgs {
    const float* const* wbase;  // [K] online parameter bases
    WtLayer layer[2];           // blockIdx.z selects the layer
    int K;
};
__device__ __forceinline__ void wt_build_body(const WtBuildArgs& args, int bx, int k, int layer) {
    const WtLayer& a = args.layer[layer];
    const long per_var = (long)a.KHs * a.KWs * a.CO * a.CI;
    const long e = (long)bx * 256 + threadIdx.x;
    if (e >= per_var * a.n_var) return;
    const int vi = (int)(e / per_var);
    long r = e - vi * per_var;
    const int ci = (int)(r % a.CI);
    r /= a.CI;
    const int co = (int)(r % a.CO);
    r /= a.CO;
    const int kws = (int)(r % a.KWs), khs = (int)(r / a.KWs);
    // variant = (row parity rh, col parity rw) of the OUTPUT position (one variant when S == 1);
    // ph = (r + PL) % S is the residue of the kernel taps that reach it, taken in descending order
    const int rh = vi / a.S, rw = vi % a.S;
    const int kh = (rh + a.PLh) % a.S + a.S * (a.KHs - 1 - khs);
    const int kw = (rw + a.PLw) % a.S + a.S * (a.KWs - 1 - kws);
    a.wt[(long)k * a.wt_stride + e] = args.wbase[k][a.w_off + ((long)(kh * a.KW + kw) * a.CI + ci) * a.CO + co];
}
__global__ __launch_bounds__(256) void k_wt_build(WtBuildArgs args) { wt_build_body(args, blockIdx.x, blockIdx.y, blockIdx.z); }

// --------------------------------------------------------------------------------------------
// Dense_0 forward, split-K partials (architectures/dqn.py:67-68).  M = 32 samples, so this is a
// weight-streaming kernel: every W element is used once per net; 16 B per lane straight to VGPRs.
// item = (net, batch block, k-split, 128-wide column tile); partials are reduced (with bias + ReLU)
// by k_head in fixed split order.
// --------------------------------------------------------------------------------------------
struct DenseFwdArgs {
    const float* in;  // [n_nets][nb][F*32]
    float* part;      // [n_nets][nb][NS][J][32]
    const float* const* wbase;
    long w_off, n_items;
    int n_nets, nb, NS, n_jt, F, J;
    int net_rot;  // work item n covers net (n + net_rot) % n_nets: the training set runs its target nets first, so that
                  // the online Dense_0 kernel is the most recently streamed 79 MB when the backward pass re-reads it
    int bb_inner;  // k_dense0_fwd3: the sample block is the FASTEST index of the work item (B > 32)
    int nt_from;   // k_dense0_fwd3: W of nets [0, nt_from) with default-policy loads, of the others non-temporally
};

__global__ __launch_bounds__(256) void k_dense0_fwd(DenseFwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bl = lane & 31, h = lane >> 5;
    long item = (long)blockIdx.x * 4 + wave;
    if (item >= a.n_items) return;
    const int jt = (int)(item % a.n_jt);
    item /= a.n_jt;
    const int s = (int)(item % a.NS);
    item /= a.NS;
    const int bb = (int)(item % a.nb);
    const int n = ((int)(item / a.nb) + a.net_rot) % a.n_nets;
    // balanced split-K: the F / 32 row units are dealt as evenly as possible (the first F/32 % NS splits get one more)
    const int units = a.F / 32, ub = units / a.NS, ur = units - ub * a.NS;
    const int f0 = 32 * (s * ub + min(s, ur)), f1 = 32 * ((s + 1) * ub + min(s + 1, ur));
    const float* W = a.wbase[n] + a.w_off + (long)(f0 + h) * a.J + jt * 128 + 4 * bl;
    const float* X = a.in + ((long)n * a.nb + bb) * a.F * 32 + (long)f0 * 32 + lane;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const long wstep = 2L * a.J;
    constexpr int U = 8;  // k-steps per chunk; (f1 - f0) is a multiple of 2U rows by construction
    const int NC = (f1 - f0) / (2 * U);
    float4 wv[2][U];
    float xv[2][U];
#define D0F_LOAD(c, s)                                                                         \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                            \
        wv[s][u] = *reinterpret_cast<const float4*>(W + ((long)(c) * U + u) * wstep);          \
        xv[s][u] = X[((long)(c) * U + u) * 64];                                                \
    }
#define D0F_MMA(s)                                                  \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                 \
        acc[0] = mfma32(wv[s][u].x, xv[s][u], acc[0]);              \
        acc[1] = mfma32(wv[s][u].y, xv[s][u], acc[1]);              \
        acc[2] = mfma32(wv[s][u].z, xv[s][u], acc[2]);              \
        acc[3] = mfma32(wv[s][u].w, xv[s][u], acc[3]);              \
    }
    D0F_LOAD(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 0; c < NC; c += 2) {  // NC >= 1 (splits are whole multiples of 2U rows); an odd tail is guarded
        D0F_LOAD(min(c + 1, NC - 1), 1)
        __builtin_amdgcn_sched_barrier(0);
        D0F_MMA(0)
        __builtin_amdgcn_sched_barrier(0);
        D0F_LOAD(min(c + 2, NC - 1), 0)
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NC) D0F_MMA(1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef D0F_LOAD
#undef D0F_MMA
    float* P = a.part + ((((long)n * a.nb + bb) * a.NS + s) * a.J + jt * 128) * 32 + bl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int i = mfma_row(r, h);
        P[(4 * i + 0) * 32] = acc[0][r];
        P[(4 * i + 1) * 32] = acc[1][r];
        P[(4 * i + 2) * 32] = acc[2][r];
        P[(4 * i + 3) * 32] = acc[3][r];
    }
}

// The same kernel on the bf16 matrix cores (convp.h arithmetic): W and the activations are split into three bf16 planes
// in registers as they arrive (W is streamed once per net and step, so a packed copy would only add HBM traffic) and
// every 16-row k-step costs 6 x 32 MFMA cycles per tile instead of 8 x 64.  The f32 version was bound by its serial
// MFMA chain (704 x 64 cycles per wave on < 1 wave per SIMD: 4.1 TB/s); this one by HBM and the split's VALU work.
__device__ __forceinline__ bf16x8 planes8(const unsigned (&p)[4]) { return __builtin_bit_cast(bf16x8, (u32x4){p[0], p[1], p[2], p[3]}); }

template <bool PLAIN = true, bool WNT = (D0_FWD_NT != 0)>  // PLAIN: the compiler's own order of split and products; WNT: non-temporal W loads
__device__ __forceinline__ void dense0_fwd3_body(const DenseFwdArgs& a) {
    constexpr int RING = 4;  // k-steps of W / activation rows in flight per lane (one wave per SIMD)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bl = lane & 31, h = lane >> 5;
    long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;  // (4 waves per workgroup; fewer when the launch has too few items to put one workgroup on every CU otherwise)
    int jt, s;
    {
        if (item >= a.n_items) return;
    }
    int bb, n;
    if (a.bb_inner) {
        // several sample blocks (B > 32): the blocks of one (net, split, column tile) are NEIGHBOURING waves -- they read the same
        // window of W at the same time, so it comes from HBM once and from L1 / L2 for the others (default-policy loads), instead of
        // once per block as with the block as the slow index (B = 256: 268 -> see profiles/r5_b256_*.txt)
        bb = (int)(item % a.nb);
        item /= a.nb;
        jt = (int)(item % a.n_jt);
        item /= a.n_jt;
        s = (int)(item % a.NS);
        n = ((int)(item / a.NS) + a.net_rot) % a.n_nets;
    } else {
        jt = (int)(item % a.n_jt);
        item /= a.n_jt;
        s = (int)(item % a.NS);
        item /= a.NS;
        bb = (int)(item % a.nb);
        n = ((int)(item / a.nb) + a.net_rot) % a.n_nets;
    }
    // INTERLEAVED split-K: split s takes the 16-row k-steps s, s + NS, s + 2 NS, ...  The NS workgroups of a net then
    // read one contiguous NS x 32 KB window of the kernel at any time, like a grid-stride copy does (6.1-6.4 TB/s for a
    // pure read, tools/probes/read_bw_probe.hip); with a contiguous row range per split the chip ran 250 separate
    // streams, each in its own DRAM pages, and topped out at 3.7-4.1 TB/s.
    const int NU = a.F / 16, NC = (NU - s + a.NS - 1) / a.NS;  // k-steps of this split (NU >= NS)
    const long step_rows = 16L * a.NS;
    // lane (bl, h): rows 16 (s + NS c) + 8 h + jj (jj = 0..7 = the MFMA's k index), W columns jt * 128 + 4 bl .. + 3 (one per
    // tile q), activation column bl
    const float* W = a.wbase[n] + a.w_off + (long)(16 * s + 8 * h) * a.J + jt * 128 + 4 * bl;
    const float* X = a.in + ((long)n * a.nb + bb) * a.F * 32 + (long)(16 * s + 8 * h) * 32 + bl;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    // a ring of four k-steps of W / activation rows per lane (32 KB of loads in flight per wave: with less than one wave
    // per SIMD the kernel was bound by how many bytes it kept in flight, not by HBM or the matrix cores)
    float4 wv[RING][8];
    float xv[RING][8];
#define D3_LOAD(c, s)                                                                          \
    _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                         \
        wv[s][jj] = ld4<WNT>(W + ((long)(c) * step_rows + jj) * a.J);                           \
        xv[s][jj] = X[((long)(c) * step_rows + jj) * 32];                                      \
    }
#define D3_GETX(s)                                                                             \
    float xr[8];                                                                               \
    _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) xr[jj] = xv[s][jj];
// One k-step: the three planes of the activations, then per column tile q six products.  hipcc leaves each tile's six MFMAs back
// to back behind its 44-instruction split (rocprofv3 --pmc, profiles/r5_d0fwd_pmc_vs_reader.txt: a wave spent 33 % of its cycles
// issuing VALU and another 32 % stalled at MFMA issue, against 9 % + 1 % for a plain reader of the same bytes in the same
// shape), so the split of tile q + 1 is threaded by hand into the gaps between the products of tile q (one operand pair per gap,
// order pinned by sched_barrier): the matrix pipe runs under the vector work instead of after it.
#define D3_PAIR(P, s, m, comp) split3_pk(wv[s][2 * (m)].comp, wv[s][2 * (m) + 1].comp, P##0[m], P##1[m], P##2[m]);
#define D3_SB __builtin_amdgcn_sched_barrier(0);
#define D3_TILE(s, q, A, B, ncomp, HAS_NEXT)                                                   \
    {                                                                                          \
        const bf16x8 w0 = planes8(A##0), w1 = planes8(A##1), w2 = planes8(A##2);               \
        D3_SB acc[q] = mfma_bf16(w2, x0, acc[q]); D3_SB                                        \
        if (HAS_NEXT) { D3_PAIR(B, s, 0, ncomp) } D3_SB                                        \
        acc[q] = mfma_bf16(w0, x2, acc[q]); D3_SB                                              \
        if (HAS_NEXT) { D3_PAIR(B, s, 1, ncomp) } D3_SB                                        \
        acc[q] = mfma_bf16(w1, x1, acc[q]); D3_SB                                              \
        if (HAS_NEXT) { D3_PAIR(B, s, 2, ncomp) } D3_SB                                        \
        acc[q] = mfma_bf16(w1, x0, acc[q]); D3_SB                                              \
        if (HAS_NEXT) { D3_PAIR(B, s, 3, ncomp) } D3_SB                                        \
        acc[q] = mfma_bf16(w0, x1, acc[q]);                                                    \
        acc[q] = mfma_bf16(w0, x0, acc[q]); D3_SB                                              \
    }
#define D3_TILE_P(s, q, comp)                                                                  \
    {                                                                                          \
        unsigned p0[4], p1[4], p2[4];                                                          \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) split3_pk(wv[s][2 * m].comp, wv[s][2 * m + 1].comp, p0[m], p1[m], p2[m]); \
        const bf16x8 w0 = planes8(p0), w1 = planes8(p1), w2 = planes8(p2);                     \
        acc[q] = mfma_bf16(w2, x0, acc[q]);                                                    \
        acc[q] = mfma_bf16(w0, x2, acc[q]);                                                    \
        acc[q] = mfma_bf16(w1, x1, acc[q]);                                                    \
        acc[q] = mfma_bf16(w1, x0, acc[q]);                                                    \
        acc[q] = mfma_bf16(w0, x1, acc[q]);                                                    \
        acc[q] = mfma_bf16(w0, x0, acc[q]);                                                    \
    }
#define D3_MMA_P(s)                                                                            \
    {                                                                                          \
        unsigned q0[4], q1[4], q2[4];                                                          \
        D3_GETX(s)                                                                             \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) split3_pk(xr[2 * m], xr[2 * m + 1], q0[m], q1[m], q2[m]); \
        const bf16x8 x0 = planes8(q0), x1 = planes8(q1), x2 = planes8(q2);                     \
        D3_TILE_P(s, 0, x) D3_TILE_P(s, 1, y) D3_TILE_P(s, 2, z) D3_TILE_P(s, 3, w)            \
    }
#define D3_MMA(s)                                                                              \
    {                                                                                          \
        unsigned q0[4], q1[4], q2[4], pa0[4], pa1[4], pa2[4], pb0[4], pb1[4], pb2[4];          \
        D3_GETX(s)                                                                             \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) split3_pk(xr[2 * m], xr[2 * m + 1], q0[m], q1[m], q2[m]); \
        const bf16x8 x0 = planes8(q0), x1 = planes8(q1), x2 = planes8(q2);                     \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) { D3_PAIR(pa, s, m, x) }                 \
        D3_TILE(s, 0, pa, pb, y, true) D3_TILE(s, 1, pb, pa, z, true)                          \
        D3_TILE(s, 2, pa, pb, w, true) D3_TILE(s, 3, pb, pa, x, false)                         \
    }
#define D3_STEP(u)                                                                             \
    D3_LOAD(min(c + (u) + RING - 1, NC - 1), ((u) + RING - 1) % RING)                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (c + (u) < NC) { if (PLAIN) D3_MMA_P(u) else D3_MMA(u) }                                \
    __builtin_amdgcn_sched_barrier(0);
    D3_LOAD(0, 0)
    D3_LOAD(min(1, NC - 1), 1)
    D3_LOAD(min(2, NC - 1), 2)
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 0; c < NC; c += RING) {
        D3_STEP(0) D3_STEP(1) D3_STEP(2) D3_STEP(3)
    }
#undef D3_STEP
#undef D3_LOAD
#undef D3_TILE
#undef D3_PAIR
#undef D3_SB
#undef D3_TILE_P
#undef D3_MMA_P
#undef D3_GETX
#undef D3_MMA
    float* P = a.part + ((((long)n * a.nb + bb) * a.NS + s) * a.J + jt * 128) * 32 + bl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int i = mfma_row(r, h);
        P[(4 * i + 0) * 32] = acc[0][r];
        P[(4 * i + 1) * 32] = acc[1][r];
        P[(4 * i + 2) * 32] = acc[2][r];
        P[(4 * i + 3) * 32] = acc[3][r];
    }
}
// Nets [0, nt_from) are read with default-policy loads, the others non-temporally.  The training set's online nets (the first K) are
// read again 50 us later by the fused update: allocated in the memory-side cache by this launch, part of those 79 MB is still there
// (update -3 ... -5 us); the target nets' 79 MB are read once per step and would only push the conv launches' working set out
// (all 158 MB default-policy: every later conv launch +0.3 ... 1 us; profiles/r5_d0_keep_online_ab.txt).  A workgroup's waves are the
// column tiles of one (net, split): the choice is workgroup-uniform.
__global__ __launch_bounds__(256) void k_dense0_fwd3(DenseFwdArgs a) {
    if (a.nt_from > 0 && !a.bb_inner) {
        const long item = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) / ((long)a.n_jt * a.NS * a.nb);
        const int n = __builtin_amdgcn_readfirstlane(((int)item + a.net_rot) % a.n_nets);
        if (n < a.nt_from) {
            dense0_fwd3_body<true, false>(a);
            return;
        }
    }
    dense0_fwd3_body<true>(a);
}
// several sample blocks per net: block-inner work items, default-policy W loads (the neighbours' re-reads hit on-chip)
// -- and the threaded split: with every window of W serving several blocks the waves are bound by their own issue, not by the stream
__global__ __launch_bounds__(256) void k_dense0_fwd3b(DenseFwdArgs a) { dense0_fwd3_body<false, false>(a); }


// --------------------------------------------------------------------------------------------
// Head, stage 1 (all 2K nets in parallel): split-K reduce + bias + ReLU -> h, and the per-chunk partial
// products of Dense_1.  grid = (J / 32 chunks, net * batch block).   architectures/dqn.py:67-70
// --------------------------------------------------------------------------------------------
struct HiddenArgs {
    const float* part;          // [2K][nb][NS][J][32]
    const float* const* wbase;  // [2K]
    long b0_off, w1_off;
    int nb, NS, J, A;
    float* hbuf;   // [2K][nb][J][32]      relu(Dense_0)
    float* qpart;  // [2K][nb][J/32][32][32]   sum over the chunk's 32 hidden units of h * W1
};

__global__ __launch_bounds__(256) void k_hidden(HiddenArgs a) {
    __shared__ float hs[32][33];
    __shared__ float w1s[32 * 32];  // this chunk's rows of W1 ([32 hidden units][A]), requested before the partials
    const int t = threadIdx.x;
    const int jc = blockIdx.x, slot = blockIdx.y;
    const float* p = a.wbase[slot / a.nb];
    const float* part = a.part + (long)slot * a.NS * a.J * 32;
    float* hb = a.hbuf + (long)slot * a.J * 32;
    {
        const float* w1 = p + a.w1_off + (long)jc * 32 * a.A;
        float wv[4];  // A <= 32: at most four elements per thread
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[r] = t + 256 * r < 32 * a.A ? w1[t + 256 * r] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (t + 256 * r < 32 * a.A) w1s[t + 256 * r] = wv[r];
    }
    // split-K reduce: thread = (row jl = t / 8, samples 4 (t % 8) .. + 3): a wave-instruction moves 8 whole rows (1 KB); the
    // partials are added in split order (reproducible), 16 splits per round in flight
    const int jl = t >> 3, l8 = t & 7, j = jc * 32 + jl;
    const float* pr = part + (long)j * 32 + 4 * l8;
    const long sstride = (long)a.J * 32;
    const float bias = p[a.b0_off + j];
    float4 sv = make_float4(bias, bias, bias, bias);
    for (int sp = 0; sp < a.NS; sp += 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            v[u] = sp + u < a.NS ? *reinterpret_cast<const float4*>(pr + (sp + u) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (sp + u < a.NS) { sv.x += v[u].x; sv.y += v[u].y; sv.z += v[u].z; sv.w += v[u].w; }
    }
    sv.x = fmaxf(sv.x, 0.f); sv.y = fmaxf(sv.y, 0.f); sv.z = fmaxf(sv.z, 0.f); sv.w = fmaxf(sv.w, 0.f);
    hs[jl][4 * l8 + 0] = sv.x; hs[jl][4 * l8 + 1] = sv.y; hs[jl][4 * l8 + 2] = sv.z; hs[jl][4 * l8 + 3] = sv.w;
    *reinterpret_cast<float4*>(hb + (long)j * 32 + 4 * l8) = sv;
    __syncthreads();
    // Dense_1 partial of this chunk: q[ac][b] = sum over its 32 hidden units (in order) of h[jl][b] * W1[jl][ac]
    const int b = t & 31, jj = t >> 5;
    for (int ac = jj; ac < a.A; ac += 8) {
        float hv[32], wv[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) { hv[r] = hs[r][b]; wv[r] = w1s[r * a.A + ac]; }
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s = fmaf(hv[r], wv[r], s);
        a.qpart[(((long)slot * (a.J / 32) + jc) * 32 + ac) * 32 + b] = s;
    }
}

// --------------------------------------------------------------------------------------------
// Head, stage 2: Q = b1 + sum of chunk partials, TD target with the wavefront max over actions, squared
// loss, dL/dq, then for this block's 32 hidden units: dL/dh (ReLU mask), Dense_0-bias and Dense_1
// gradients.  grid = (J / 32 chunks, head); every block re-derives the (tiny) TD part.
//   idqn.py:111-124  loss_on_batch / loss / compute_target
// --------------------------------------------------------------------------------------------
struct TdArgs {
    const float* hbuf;
    const float* qpart;
    const float* const* wbase;
    long b0_off, w1_off, b1_off, P;
    long gP, g_b0_off, g_w1_off, g_b1_off;  // gradient arena: per-head stride and leaf offsets (see GradLayout)
    int K, nb, J, A, B, Bdiv;
    const int32_t* action;
    const float* reward;
    const uint8_t* terminal;
    float gamma_n;
    float* dh;      // [K][nb][J][32]
    float* q_dbg;   // [2K][nb][32][32]
    float* grad;    // [K][P]
    float* losses;  // [K]
    int32_t* count;        // [K] optax step counter (pre-increment)
    float* bcinv;          // [K][2] out: reciprocal Adam bias corrections of THIS step
    float b1, b2;          // Adam decay rates (f32, as folded by the host)
    double* cum;           // [K] running f64 sum of the per-head losses (idqn.py:72)
    int finish_step;       // 1: this launch also does count += 1 and cum += loss (every later kernel of the step
                           //    reads bcinv, not count); 0: two-phase step, idqn_apply_adam's epilogue does it
    int bcinv_done;          // 1: the staging launch of this step already wrote bcinv (plane path)
    const float* is_weight;  // [B] per-sample loss weights (prioritized-replay extension) or nullptr = plain mean
    float* td_abs;           // [K][B] out: |TD error| per head and sample, or nullptr
    long long* prof;         // debug (IDQN_CONV_PROF=9): 8 stamps per workgroup, or nullptr
    // Several sample blocks, one workgroup per (chunk, head, BLOCK) (gridDim.z = nb) instead of a loop over the blocks: every
    // workgroup leaves its block's gradient partials in bpart[(head, chunk)][block][TD_BPART] and adds an arrival; the last one
    // to arrive adds the nb partials IN BLOCK ORDER -- the loop's order, bit-identical -- and writes the leaves.  nullptr: the loop.
    float* bpart;
    unsigned* bctr;          // [K * J / 32], zero between launches
};
constexpr int TD_BPART = 1152;  // 1024 Dense_1 kernel elements of the chunk, 32 Dense_0 biases, 32 Dense_1 biases, the loss

__device__ __forceinline__ void td_dh_body(const TdArgs& a, int jc, int k, const int bb_only = -1) {
    __shared__ float hs[32][33];
    __shared__ float qo[32 * 32], qt[32 * 32];
    __shared__ float qmax[32], cs[32], red[1];
    __shared__ int acts[32];
    __shared__ float w1s[32 * 32];  // this chunk's rows of W1 ([32 hidden units][A]), loaded while the q partials arrive
    const int t = threadIdx.x, lane = t & 63, bl = lane & 31, h = lane >> 5;
    const int b = t & 31, jj = t >> 5, NJC = a.J / 32;
    const float* po = a.wbase[k];
    const float* pt = a.wbase[a.K + k];
    const float* w1 = po + a.w1_off;
    float* G = a.grad + (long)k * a.gP;
    float gw[4] = {0.f, 0.f, 0.f, 0.f}, gb0[4] = {0.f, 0.f, 0.f, 0.f}, gb1 = 0.f, loss_acc = 0.f;
    long long ts[8];
    ts[0] = clock64();
    for (int bb = bb_only < 0 ? 0 : bb_only; bb < (bb_only < 0 ? a.nb : bb_only + 1); ++bb) {
        const long so = (long)k * a.nb + bb, st = (long)(a.K + k) * a.nb + bb;
        // this block's hidden rows do not depend on the q reduction below: have them in flight meanwhile
        const float* hb = a.hbuf + so * a.J * 32 + (long)jc * 32 * 32;
        float hreg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hreg[i] = hb[(jj + 8 * i) * 32 + b];
        // nor do the batch's scalars (wave 0 consumes them after the reduction) and this chunk's 32 rows of W1
        int s_act = 0;
        float s_rew = 0.f, s_wgt = 1.0f;
        unsigned s_term = 0;
        const int bg0 = bb * 32 + bl;
        if (t < 64 && bg0 < a.B) {
            s_act = a.action[bg0];
            s_rew = a.reward[bg0];
            s_term = a.terminal[bg0];
            if (a.is_weight) s_wgt = a.is_weight[bg0];
        }
        // (this chunk's 32 rows of W1 -- A <= 32: at most four elements per thread -- and the Dense_1 biases hang on the nets' base
        // pointers, which are themselves loaded (wbase[k]): they are requested BEHIND the q partials, which need nothing but the
        // kernel arguments, so that the pointer fetch is not a round trip of its own in front of everything else)
        float wv[4] = {0.f, 0.f, 0.f, 0.f};
        // (action, sample) elements in rounds of 256: every partial of every round of this thread (NJC = J / 32 <= 16 chunks x
        // 2 nets x up to 4 rounds) is requested before the first add -- one load latency whatever A is
        auto q_reduce = [&](auto NR_) {
            constexpr int NR = decltype(NR_)::value;
            float x[NR][16], y[NR][16], b1o[NR], b1t[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int e = min(t + 256 * r, a.A * 32 - 1);
                const float* qo_ = a.qpart + so * NJC * 1024 + e;
                const float* qt_ = a.qpart + st * NJC * 1024 + e;
#pragma unroll
                for (int u = 0; u < 16; ++u) {  // (unconditional loads, clamped: as `u < NJC ? load : 0` every load sat in a branch of
                    const int uc = u < NJC ? u : 0;  //  its own and hipcc waited for the first pair -- vmcnt(0) -- before the other 30 went out)
                    x[r][u] = qo_[uc * 1024];
                    y[r][u] = qt_[uc * 1024];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int e = min(t + 256 * r, a.A * 32 - 1);
                b1o[r] = po[a.b1_off + (e >> 5)];
                b1t[r] = pt[a.b1_off + (e >> 5)];
            }
            if (bb == 0 || bb_only >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) wv[r] = t + 256 * r < 32 * a.A ? w1[(long)jc * 32 * a.A + t + 256 * r] : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int e = t + 256 * r;
                if (e >= a.A * 32) break;
                const int ac = e >> 5;
                float vo = 0.f, vt = 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (u < NJC) { vo += x[r][u]; vt += y[r][u]; }
                vo += b1o[r];
                vt += b1t[r];
                qo[e] = vo;
                qt[e] = vt;
                if (jc == 0) {
                    a.q_dbg[so * 1024 + e] = vo;
                    a.q_dbg[st * 1024 + e] = vt;
                }
            }
        };
        switch ((a.A * 32 + 255) / 256) {  // A <= 32 (idqn_create)
            case 1: q_reduce(std::integral_constant<int, 1>{}); break;
            case 2: q_reduce(std::integral_constant<int, 2>{}); break;
            case 3: q_reduce(std::integral_constant<int, 3>{}); break;
            default: q_reduce(std::integral_constant<int, 4>{}); break;
        }
        if (bb == 0 || bb_only >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (t + 256 * r < 32 * a.A) w1s[t + 256 * r] = wv[r];
        }
        __syncthreads();
        ts[1] = clock64();
        if (t < 64) {  // wave 0: max over actions -- each half-wave folds every other action, one cross-lane step
            float m = -INFINITY;
            for (int ac = h; ac < a.A; ac += 2) m = fmaxf(m, qt[ac * 32 + bl]);
            m = fmaxf(m, __shfl_xor(m, 32));
            const int bg = bg0;
            const bool valid = bg < a.B;
            const int ac = valid ? s_act : 0;
            float td = 0.f;
            if (valid) {
                // idqn.py:122  r + (1 - terminal) * gamma**n * max_a Q_target(s')
                const float tgt = s_rew + (float)(1 - (int)s_term) * a.gamma_n * m;
                td = qo[ac * 32 + bl] - tgt;
            }
            const float wgt = valid ? s_wgt : 1.0f;
            if (h == 0) {
                cs[bl] = 2.0f * wgt * td / (float)a.Bdiv;
                acts[bl] = ac;
                if (jc == 0 && valid && a.td_abs) a.td_abs[(long)k * a.B + bg] = fabsf(td);
            }
            float sq = (h == 0) ? wgt * td * td : 0.f;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
            if (lane == 0) red[0] = sq;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) hs[jj + 8 * i][b] = hreg[i];
        __syncthreads();
        ts[2] = clock64();
        loss_acc += red[0];
        float* dh = a.dh + so * a.J * 32 + (long)jc * 32 * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int jl = jj + 8 * i;
            float d = hs[jl][b] > 0.f ? w1s[jl * a.A + acts[b]] * cs[b] : 0.f;
            dh[jl * 32 + b] = d;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) d += __shfl_xor(d, o);
            gb0[i] += d;
        }
        ts[3] = clock64();
        // Dense_1 weight gradient: gw[jl][ac] = sum over the samples x with action ac of h[jl][x] * cs[x] (sample order).
        // Every LDS operand is read into registers first -- written as `cond ? hs[..] * cs[..] : 0` hipcc guarded the two
        // reads with a branch per sample: 170 cycles per sample, 2.6 us of the kernel at A = 6, 7.7 us at A = 18.
        int av[32];
        float cv[32];
#pragma unroll
        for (int x = 0; x < 32; ++x) { av[x] = acts[x]; cv[x] = cs[x]; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int o = t + 256 * m;
            if (o < 32 * a.A) {
                const int jl = o / a.A, ac = o - jl * a.A;
                float hv[32];
#pragma unroll
                for (int x = 0; x < 32; ++x) hv[x] = hs[jl][x];
                float s = 0.f;
#pragma unroll
                for (int x = 0; x < 32; ++x) {
                    const float pr = hv[x] * cv[x];
                    s += av[x] == ac ? pr : 0.f;
                }
                gw[m] += s;
            }
        }
        ts[4] = clock64();
        if (jc == 0 && t < a.A) {
            float s = 0.f;
#pragma unroll
            for (int x = 0; x < 32; ++x) s += av[x] == t ? cv[x] : 0.f;
            gb1 += s;
        }
        __syncthreads();
    }
    if (bb_only >= 0) {
        // this block's partials past the non-coherent caches, then the arrival; the last workgroup of the (head, chunk) goes on
        float* P = a.bpart + (((long)k * NJC + jc) * a.nb + bb_only) * TD_BPART;
#pragma unroll
        for (int m = 0; m < 4; ++m) __hip_atomic_store(P + t + 256 * m, gw[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (b == 0) __hip_atomic_store(P + 1024 + jj + 8 * i, gb0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < 32) __hip_atomic_store(P + 1056 + t, gb1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0) __hip_atomic_store(P + 1088, loss_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            unsigned* ctr = a.bctr + (long)k * NJC + jc;
            const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)a.nb - 1u;
            if (last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed
            acts[0] = last;
        }
        __syncthreads();
        if (!acts[0]) return;
        const float* P0 = a.bpart + ((long)k * NJC + jc) * a.nb * TD_BPART;
#pragma unroll
        for (int m = 0; m < 4; ++m) gw[m] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) gb0[i] = 0.f;
        gb1 = 0.f;
        loss_acc = 0.f;
        for (int x0 = 0; x0 < a.nb; x0 += 8) {  // (the loop's order: 0 + p0 + p1 + ...; eight blocks' partials in flight at once)
            float pw[8][4], pb0[8][4], pb1[8], pl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* Px = P0 + (long)min(x0 + u, a.nb - 1) * TD_BPART;
#pragma unroll
                for (int m = 0; m < 4; ++m) pw[u][m] = __hip_atomic_load(Px + t + 256 * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < 4; ++i) pb0[u][i] = __hip_atomic_load(Px + 1024 + jj + 8 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pb1[u] = __hip_atomic_load(Px + 1056 + (t & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pl[u] = __hip_atomic_load(Px + 1088, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (x0 + u < a.nb) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) gw[m] += pw[u][m];
#pragma unroll
                    for (int i = 0; i < 4; ++i) gb0[i] += pb0[u][i];
                    gb1 += pb1[u];
                    loss_acc += pl[u];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (b == 0) G[a.g_b0_off + jc * 32 + jj + 8 * i] = gb0[i];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int o = t + 256 * m;
        if (o < 32 * a.A) G[a.g_w1_off + (long)jc * 32 * a.A + o] = gw[m];
    }
    if (a.prof && t == 0) {
        ts[5] = clock64();
        long long* pr = a.prof + ((long)k * NJC + jc) * 8;
        for (int i = 0; i < 6; ++i) pr[i] = ts[i];
    }
    if (jc == 0) {
        if (t < a.A) G[a.g_b1_off + t] = gb1;
        if (t == 0) {
            a.losses[k] = loss_acc / (float)a.Bdiv;
            if (!a.bcinv_done) {
                const double tt = (double)(a.count[k] + 1);
                a.bcinv[2 * k] = 1.0f / (1.0f - (float)pow((double)a.b1, tt));
                a.bcinv[2 * k + 1] = 1.0f / (1.0f - (float)pow((double)a.b2, tt));
            }
            if (a.finish_step) {
                a.count[k] += 1;
                a.cum[k] = a.cum[k] + (double)(loss_acc / (float)a.Bdiv);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_td_dh(TdArgs a) {
    warm_kernargs<sizeof(TdArgs)>();
    td_dh_body(a, blockIdx.x, blockIdx.y, a.bpart ? (int)blockIdx.z : -1);
}

// The TD / loss kernel has J / 32 x K workgroups (80 for the Atari net) and is latency-bound; the re-indexing of the
// Conv_1 / Conv_2 kernels for the data gradients (k_wt_build) depends on nothing this step computes, so its
// workgroups ride in the same launch instead of costing one of their own.
__global__ __launch_bounds__(256) void k_td_dh_wt(TdArgs a, WtBuildArgs w, int wt_nx) {
    warm_kernargs<(sizeof(TdArgs) + sizeof(WtBuildArgs) + 16 < 2048 ? sizeof(TdArgs) + sizeof(WtBuildArgs) + 16 : 2048)>();
    const int njc = a.J / 32, n_td = njc * a.K, b = blockIdx.x;
    if (b < n_td) {
        td_dh_body(a, b % njc, b / njc);
    } else {
        const int r = b - n_td, per_layer = wt_nx * w.K;
        wt_build_body(w, r % wt_nx, (r / wt_nx) % w.K, r / per_layer);
    }
}

// Inference head after k_hidden: q[b][a] = b1[a] + sum over the J / 32 chunk partials, and the greedy action (first
// maximum, jnp.argmax) of every state.  One small workgroup; replaces the single-workgroup k_head_q (345 us) on the
// acting path.
struct QOutArgs {
    const float* qpart;         // [1][1][J/32][32][32]
    const float* const* wbase;  // [1]
    long b1_off;
    int NJC, A, n;
    float* q_out;     // [n][A]
    int32_t* action;  // [n] or nullptr
};
__global__ __launch_bounds__(256) void k_q_out(QOutArgs a) {
    __shared__ float qs[32 * 32];
    const float* p = a.wbase[0];
    for (int e = threadIdx.x; e < a.A * 32; e += 256) {
        const int ac = e >> 5, b = e & 31;
        float v = 0.f;
        for (int c = 0; c < a.NJC; c += 4) {  // NJC is a multiple of 4; added in chunk order like k_td_dh
            const float x0 = a.qpart[(c + 0) * 1024 + e], x1 = a.qpart[(c + 1) * 1024 + e];
            const float x2 = a.qpart[(c + 2) * 1024 + e], x3 = a.qpart[(c + 3) * 1024 + e];
            v = (((v + x0) + x1) + x2) + x3;
        }
        v += p[a.b1_off + ac];
        qs[e] = v;
        if (b < a.n) a.q_out[b * a.A + ac] = v;
    }
    __syncthreads();
    if (a.action && (int)threadIdx.x < a.n) {
        const int b = threadIdx.x;
        int best = 0;
        float bv = qs[b];
        for (int ac = 1; ac < a.A; ++ac)
            if (qs[ac * 32 + b] > bv) { bv = qs[ac * 32 + b]; best = ac; }
        a.action[b] = best;
    }
}

// --------------------------------------------------------------------------------------------
// Dense_0 data gradient: da3[f][b] = relu'(a3[f][b]) * sum_j W0[f][j] * dh[j][b]
// workgroup = (head, batch block, 4 consecutive 32-row f tiles).  W rows are read 16 floats per lane.
// --------------------------------------------------------------------------------------------
struct DenseDgradArgs {
    const float* dh;   // [K][nb][J][32]
    const float* a3;   // [2K][nb][F*32]  (online nets first)
    float* da3;        // [K][nb][g.block] f32 rows or nullptr (plane path)
    unsigned short* da3p;  // the same pixels as three bf16 planes (convp.h layout, geometry g) or nullptr
    float* pb;             // [K * nb][H * W][C] sums of da3 over the 32 samples (Conv_2 bias gradient) or nullptr
    const float* const* wbase;
    long w_off, n_items;
    int K, nb, n_ft, F, J, C;  // C = channels of a3 (f = pos*C + c)
    ActGeom g;                 // geometry of da3
    float* raw;                // i-IQN: plain rows [K][nb][F][32] of W0 . dh, no mask, none of the outputs above (or nullptr)
};

template <int WAVES>  // f tiles (= waves) per workgroup: 4, or 3 when that fills the chip more evenly
__global__ __launch_bounds__(64 * WAVES) void k_dense0_dgrad(DenseDgradArgs a) {
    // workgroup = WAVES consecutive f tiles of one (head, batch block).  dh (J x 32, <= 64 KB) is staged once
    // by LDS-DMA and shared by the waves (A operand, rows = samples: conflict-free ds_read_b32); the W rows (B operand,
    // columns = the tile's 32 f rows, 16 floats per lane and chunk) stream from HBM with register double buffering.
    // D[b][f]: a lane ends up with ONE row f and 4 x 4 consecutive samples -- float4 mask loads / stores, 8-byte plane
    // pieces, and the sum over the samples is 15 adds and one cross-half shuffle.
    extern __shared__ __attribute__((aligned(16))) float dlds[];  // [J][32]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, bl = lane & 31, h = lane >> 5;
    const int n_wg_ft = (a.n_ft + WAVES - 1) / WAVES;
    int item = xcd_contiguous_id();  // an XCD keeps to (mostly) one head: its dh stays in that L2
    const int fg = item % n_wg_ft;
    item /= n_wg_ft;
    const int bb = item % a.nb;
    const int k = item / a.nb;
    const float* dsrc = a.dh + ((long)k * a.nb + bb) * a.J * 32;
    for (int o = 0; o < a.J * 32; o += 256 * WAVES)
        if (o + wave * 256 < a.J * 32) glds16(dsrc + o + t * 4, &dlds[o + wave * 256]);  // wave-uniform guard
    const int ft = min(fg * WAVES + wave, a.n_ft - 1);
    const bool live = fg * WAVES + wave < a.n_ft;
    const int f0 = ft * 32;
    const float* W = a.wbase[k] + a.w_off + (long)(f0 + bl) * a.J + 16 * h;
    const float* D = dlds + (16 * h) * 32 + bl;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int NC = a.J / 32;  // even (J is a multiple of 128)
    float4 wv[2][4];
#define D0D_LOAD(c, s) \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) wv[s][u] = *reinterpret_cast<const float4*>(W + (c) * 32 + 4 * u);
#define D0D_MMA(c, s)                                                          \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                            \
        acc = mfma32(D[((c) * 32 + 4 * u + 0) * 32], wv[s][u].x, acc);         \
        acc = mfma32(D[((c) * 32 + 4 * u + 1) * 32], wv[s][u].y, acc);         \
        acc = mfma32(D[((c) * 32 + 4 * u + 2) * 32], wv[s][u].z, acc);         \
        acc = mfma32(D[((c) * 32 + 4 * u + 3) * 32], wv[s][u].w, acc);         \
    }
    D0D_LOAD(0, 0)
    D0D_LOAD(1, 1)
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes dh to all waves
    for (int c = 0; c < NC; c += 2) {
        D0D_MMA(c, 0)
        __builtin_amdgcn_sched_barrier(0);
        D0D_LOAD(min(c + 2, NC - 1), 0)
        __builtin_amdgcn_sched_barrier(0);
        D0D_MMA(c + 1, 1)
        __builtin_amdgcn_sched_barrier(0);
        D0D_LOAD(min(c + 3, NC - 1), 1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef D0D_LOAD
#undef D0D_MMA
    if (!live) return;
    // this lane: row f = f0 + bl, samples (r & 3) + 8 (r >> 2) + 4 h
    const int f = f0 + bl;
    if (a.raw) {
        float* O = a.raw + (((long)k * a.nb + bb) * a.F + f) * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(O + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        return;
    }
    const float* A3 = a.a3 + ((long)k * a.nb + bb) * a.F * 32 + (long)f * 32 + 4 * h;
    float4 mk[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) mk[g] = *reinterpret_cast<const float4*>(A3 + 8 * g);
    float val[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        val[4 * g + 0] = mk[g].x > 0.f ? acc[4 * g + 0] : 0.f;
        val[4 * g + 1] = mk[g].y > 0.f ? acc[4 * g + 1] : 0.f;
        val[4 * g + 2] = mk[g].z > 0.f ? acc[4 * g + 2] : 0.f;
        val[4 * g + 3] = mk[g].w > 0.f ? acc[4 * g + 3] : 0.f;
    }
    const int pos = f / a.C, c = f - pos * a.C;
    const int oh = pos / a.g.W, ow = pos - oh * a.g.W;
    const long pix = (long)(oh + a.g.lo_h) * a.g.Wp + (ow + a.g.lo_w);
    if (a.da3) {
        float* O = a.da3 + ((long)k * a.nb + bb) * a.g.block + (pix * a.C + c) * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(O + 8 * g) = make_float4(val[4 * g], val[4 * g + 1], val[4 * g + 2], val[4 * g + 3]);
    }
    if (a.da3p) {
        unsigned short* O = a.da3p + ((long)k * a.nb + bb) * a.g.block * 3 + pix * (3L * a.C * 32) + (long)c * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned q0a, q1a, q2a, q0b, q1b, q2b;
            split3_pk(val[4 * g + 0], val[4 * g + 1], q0a, q1a, q2a);
            split3_pk(val[4 * g + 2], val[4 * g + 3], q0b, q1b, q2b);
            *reinterpret_cast<uint2*>(O + 8 * g) = make_uint2(q0a, q0b);
            *reinterpret_cast<uint2*>(O + (long)a.C * 32 + 8 * g) = make_uint2(q1a, q1b);
            *reinterpret_cast<uint2*>(O + 2L * a.C * 32 + 8 * g) = make_uint2(q2a, q2b);
        }
    }
    if (a.pb) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += val[r];
        s += __shfl_xor(s, 32);
        if (h == 0) a.pb[(((long)k * a.nb + bb) * (a.g.H * a.g.W) + pos) * a.C + c] = s;
    }
}

// f32 factors of the Dense_0 gradient -> three exact bf16 planes (convp.h arithmetic), compact per (sample block, head).
// The factored data-parallel step runs the fused update over the GLOBAL batch: its MFMA work grows with the number of
// ranks while its HBM traffic does not (f32 MFMA: +16 us per extra sample block), so there the contraction runs on the
// bf16 matrix cores from planes that are split ONCE here instead of in every one of the 2420 workgroups.
struct SplitFactorsArgs {
    const float *a3, *dh;
    long a3_outer, a3_head, a3_inner, dh_outer, dh_head, dh_inner;
    int K, nb, nb_inner, F, J;
    unsigned short *a3p, *dhp;  // a3p == nullptr: dh only
};
__global__ __launch_bounds__(256) void k_split_factors(SplitFactorsArgs a) {
    const int slot = blockIdx.y, bb = slot / a.K, k = slot - bb * a.K;
    const int bo = bb / a.nb_inner, bi = bb - bo * a.nb_inner;
    const long na = (long)a.F * 4, nd = (long)a.J * 4;  // pairs of float4 per (block, head): 16-byte stores per plane
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (!a.a3p) e += na;  // dh only: the update kernel splits its a3 tiles itself (dense0_update.h, ALDS) and the grid covers J rows
    const float* src;
    unsigned short* dst;
    long plane;
    if (e < na) {
        src = a.a3 + bo * a.a3_outer + k * a.a3_head + bi * a.a3_inner;
        dst = a.a3p + (long)slot * a.F * 32;
        plane = (long)a.nb * a.K * a.F * 32;
    } else {
        e -= na;
        if (e >= nd) return;
        src = a.dh + bo * a.dh_outer + k * a.dh_head + bi * a.dh_inner;
        dst = a.dhp + (long)slot * a.J * 32;
        plane = (long)a.nb * a.K * a.J * 32;
    }
    const float4 v = *reinterpret_cast<const float4*>(src + e * 8), w = *reinterpret_cast<const float4*>(src + e * 8 + 4);
    unsigned q0[4], q1[4], q2[4];
    split3_pk(v.x, v.y, q0[0], q1[0], q2[0]);
    split3_pk(v.z, v.w, q0[1], q1[1], q2[1]);
    split3_pk(w.x, w.y, q0[2], q1[2], q2[2]);
    split3_pk(w.z, w.w, q0[3], q1[3], q2[3]);
    *reinterpret_cast<u32x4*>(dst + e * 8) = (u32x4){q0[0], q0[1], q0[2], q0[3]};
    *reinterpret_cast<u32x4*>(dst + plane + e * 8) = (u32x4){q1[0], q1[1], q1[2], q1[3]};
    *reinterpret_cast<u32x4*>(dst + 2 * plane + e * 8) = (u32x4){q2[0], q2[1], q2[2], q2[3]};
}


template <bool FUSE_ADAM, int NQ, bool FUSE_DG = false, bool BF3 = false, int RT = 1>
__global__ __launch_bounds__(256) void k_dense0_wgrad(DenseWgradArgs a) {
    __shared__ __attribute__((aligned(16))) float gs[32 * RT * 128 * NQ + (FUSE_DG ? 4096 : 0)];
    dense0_wgrad_body<FUSE_ADAM, NQ, FUSE_DG, BF3, RT>(a, (int)blockIdx.x + a.item0, gs, (int)threadIdx.x);
}
// The fused update on pairs of column tiles (dense0_update.h, dense0_pair_body): one workgroup per (head, 32 rows)
// (cache policy: TH_ST_NT = false keeps theta_new on chip; ALL_DEFAULT: one or two heads, every stream default-policy; qnet.hip d0_keep_heads.
// A template parameter, not a branch on the head: with both bodies in one kernel hipcc spilled 300 bytes per lane.)
template <bool ROWPAIR, bool TH_ST_NT = true, bool ALL_DEFAULT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_dense0_wgrad_pair(DenseWgradArgs a) {
    __shared__ __attribute__((aligned(16))) float gs[32 * 256 + 4096 + 1024];
    dense0_pair_body<ROWPAIR, 4, TH_ST_NT, ALL_DEFAULT>(a, (int)blockIdx.x + a.item0, gs, (int)threadIdx.x);
}
// The factored data-parallel update with the a3 fragments through LDS (dense0_update.h, ALDS): 48 KB of fragments before the
// 32 KB tile takes their place; three workgroups per CU like the register version (136 + 32 registers).
template <int RT, bool TH_ST_NT = true>  // 32 * RT rows x 256 columns
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT == 1 ? 3 : 2))) void k_dense0_wgrad_alds(DenseWgradArgs a) {
    __shared__ __attribute__((aligned(1024))) float gs[RT == 1 ? 32 * 256 + 4096 : 64 * 256];
    dense0_wgrad_body<true, 2, false, true, RT, true, TH_ST_NT>(a, (int)blockIdx.x + a.item0, gs, (int)threadIdx.x);
}

// Sum of the column tiles' partial data gradients, ReLU mask of a3, and the three output forms of dL/da3: bf16 planes
// (zero-bordered, the Conv_2 gradients read them), the per-position sums over the samples (Conv_2 bias gradient), f32 rows
// (f32 conv path).  One thread = 4 samples of one row f; 8 threads = a row; a wave = 8 rows = 512 contiguous plane bytes.
struct Da3FinalizeArgs {
    const float* dpart;  // [n_jt][K * nb][F][32]
    const float* a3;     // [2K][nb][F * 32] (online nets first)
    float* da3;          // f32 rows [K][nb][g.block] or nullptr
    unsigned short* da3p;
    float* pb;           // [K * nb][H * W][C] or nullptr
    long n_rows;         // K * nb * F
    int n_jt, F, C, K, nb;
    ActGeom g;
};
__global__ __launch_bounds__(256) void k_da3_finalize(Da3FinalizeArgs a) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;  // float4 index
    const long row = e >> 3;
    const int slot = (int)(e & 7);
    if (row >= a.n_rows) return;
    const long slab = a.n_rows * 32;
    float4 s = *reinterpret_cast<const float4*>(a.dpart + e * 4);
    for (int j = 1; j < a.n_jt; ++j) {
        const float4 y = *reinterpret_cast<const float4*>(a.dpart + j * slab + e * 4);
        s.x += y.x; s.y += y.y; s.z += y.z; s.w += y.w;
    }
    const float4 m = *reinterpret_cast<const float4*>(a.a3 + e * 4);  // online nets are the first K * nb slots
    s.x = m.x > 0.f ? s.x : 0.f; s.y = m.y > 0.f ? s.y : 0.f; s.z = m.z > 0.f ? s.z : 0.f; s.w = m.w > 0.f ? s.w : 0.f;
    const long sl = row / a.F;
    const int f = (int)(row - sl * a.F), pos = f / a.C, c = f - pos * a.C;
    const int oh = pos / a.g.W, ow = pos - oh * a.g.W;
    const long pix = (long)(oh + a.g.lo_h) * a.g.Wp + (ow + a.g.lo_w);
    if (a.da3) *reinterpret_cast<float4*>(a.da3 + sl * a.g.block + (pix * a.C + c) * 32 + slot * 4) = s;
    if (a.da3p) {
        unsigned short* O = a.da3p + sl * a.g.block * 3 + pix * (3L * a.C * 32) + (long)c * 32 + slot * 4;
        unsigned q0a, q1a, q2a, q0b, q1b, q2b;
        split3_pk(s.x, s.y, q0a, q1a, q2a);
        split3_pk(s.z, s.w, q0b, q1b, q2b);
        *reinterpret_cast<uint2*>(O) = make_uint2(q0a, q0b);
        *reinterpret_cast<uint2*>(O + (long)a.C * 32) = make_uint2(q1a, q1b);
        *reinterpret_cast<uint2*>(O + 2L * a.C * 32) = make_uint2(q2a, q2b);
    }
    if (a.pb) {  // sum over the row's 32 samples: 4 in this thread, then its 8 neighbours (fixed order)
        float r = (s.x + s.y) + (s.z + s.w);
        r += __shfl_xor(r, 1);
        r += __shfl_xor(r, 2);
        r += __shfl_xor(r, 4);
        if (slot == 0) a.pb[(sl * (a.g.H * a.g.W) + pos) * a.C + c] = r;
    }
}

// --------------------------------------------------------------------------------------------
// conv data gradient (+ ReLU mask of the layer below): NOT a kernel of its own.
//   din[ih][iw][ci][b] = relu'(act_in) * sum_{kh,kw,co} W[kh][kw][ci][co] * dout[oh][ow][co][b],
//   oh = (ih + PL - kh) / S for the kh with (ih + PL - kh) % S == 0
// is a stride-1 forward convolution over the zero-bordered dout buffer with flipped / transposed weights
// (per output parity when S > 1), so it runs on k_conv_fwd (epilogue 1) after k_wt_build -- see cnn_backward.
// --------------------------------------------------------------------------------------------

// --------------------------------------------------------------------------------------------
// conv weight gradient: partial slabs over chunks of output positions
//   gW[kh][kw][ci][co] = sum_{b, oh, ow} in[oh*S+kh][ow*S+kw][ci][b] * dout[oh][ow][co][b]
// workgroup = (head, kh, kw, position chunk); NIT x NOT tiles of 32x32 ([ci tile][co tile]), one per wave.
// Conv_0 (CI = 4) runs the same code with the 32 rows (kw, ci) of one kernel row as its "ci tile".
// --------------------------------------------------------------------------------------------
struct ConvWgradArgs {
    const float* in;    // [n_in_sets or 2K][nb][gin.block]  forward input of the conv (online set / nets first)
    const float* dout;  // [K][nb][gd.block]
    float* slab;        // [npc][K][slab_stride]   weights then bias
    long in_net_stride; // 0 when all heads share the input (Conv_0 reads the staged s)
    long slab_stride, n_items;
    int K, nb, npc, KH, KWe, S, in_C, CIe, CO, OH, OW, pos_per_chunk;
    ActGeom gin, gd;
};

// v3: operands through LDS.  The k index (sample) is the contiguous one of both operands, so an MFMA lane
// needs 16 floats of ITS OWN row; loading those straight from memory makes every lane touch its own cache
// line (32 lines per dwordx4 instruction: the kernel was bound by the texture-address path, MFMA util 21-26 %).
// Now the 256 threads copy whole 32-row blocks (4 KB, contiguous) with coalesced 16-B loads into an LDS image
// whose rows are padded to 36 floats, and each lane reads its row with 4 x ds_read_b128: with stride 36 the 16
// lanes of a ds_read_b128 group hit 16 distinct 4-bank slots (36 r mod 64 = 4 (9 r mod 16)) -- conflict-free.
// The 4 waves are (tile pair) x (position slot): <2,2> = 4 tile pairs x 1 position, <1,2> = 2 x 2, <1,1> = 1 x 4.
template <int NIT, int NOT>
__device__ __forceinline__ void conv_wgrad_body(const ConvWgradArgs& a, int item /* XCD-contiguous */, float* lds) {
    constexpr int NTP = NIT * NOT, PS = 4 / NTP, RPS = (NIT + NOT) * 32, ROWS = PS * RPS, NLD = ROWS / 32;
    constexpr int LDR = 36;  // padded row stride (floats); lds: 2 * ROWS * LDR floats
    static_assert(NTP == 1 || NTP == 2 || NTP == 4, "4 waves = tile pairs x position slots");
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, bl = lane & 31, h = lane >> 5;
    // workgroup = (head, chunk of output positions, kh, kw) with the TAP fastest, on an XCD-contiguous index:
    // the KH*KW workgroups that re-read one chunk's rows run back to back on one XCD (its L2 serves the re-reads;
    // with the taps spread over all XCDs every L2 fetched every row: ~9x the traffic, 1.4 us per 16 KB round)
    const int kw = item % a.KWe;
    item /= a.KWe;
    const int kh = item % a.KH;
    item /= a.KH;
    const int pc = item % a.npc;
    const int k = item / a.npc;
    const int tp = wave % NTP, slot = wave / NTP, it = tp / NOT, ot = tp % NOT;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    const int npos = a.OH * a.OW;
    const int p0 = pc * a.pos_per_chunk, p1 = min(npos, p0 + a.pos_per_chunk);
    const int nrp = (p1 - p0 + PS - 1) / PS, NR = a.nb * nrp;  // rounds: PS positions each
    const float* IN0 = a.in + (long)k * a.in_net_stride + t * 4;
    const float* DO0 = a.dout + (long)k * a.nb * a.gd.block + t * 4;
    const int wrow = (t >> 3) * LDR + (t & 7) * 4;  // where this thread's float4 of a 32-row block lands
    // Global loads run THREE rounds ahead in a register ring (a workgroup is a serial chain of rounds of only
    // ~0.5 us of MFMA work each, and few workgroups share a CU: with one round of lookahead every round paid
    // a full load latency).  The ring is 3 x 8 NAMED float4 registers: as arrays hipcc demoted it to scratch.
    const int arow = (slot * RPS + it * 32 + bl) * LDR + 16 * h;
    const int brow = (slot * RPS + NIT * 32 + ot * 32 + bl) * LDR + 16 * h;
    // source of block b (b-th 32-row block of a round: [slot][in tiles..., out tiles...]) of round e; branch-free
    // (positions clamped; a slot past the chunk end is skipped at compute time)
    auto src = [&](int e, int b) -> const float* {
        const int bb_ = e / nrp, pr_ = e - bb_ * nrp;
        const int s_ = b / (NIT + NOT), x_ = b - s_ * (NIT + NOT);
        const int pos_ = min(p0 + pr_ * PS + s_, p1 - 1);
        const int oh_ = pos_ / a.OW, ow_ = pos_ - oh_ * a.OW;
        return x_ < NIT ? IN0 + (long)bb_ * a.gin.block +
                              (((long)(oh_ * a.S + kh) * a.gin.Wp + (ow_ * a.S + kw)) * a.in_C) * 32 + x_ * 1024
                        : DO0 + (long)bb_ * a.gd.block +
                              (((long)(oh_ + a.gd.lo_h) * a.gd.Wp + (ow_ + a.gd.lo_w)) * a.CO) * 32 + (x_ - NIT) * 1024;
    };
    float4 A0, A1, A2, A3, A4, A5, A6, A7, B0, B1, B2, B3, B4, B5, B6, B7, C0, C1, C2, C3, C4, C5, C6, C7;
#define CW_L1(R, b, e) if (b < NLD) R##b = *reinterpret_cast<const float4*>(src(e, b));
#define CW_GLOAD(e, R) { CW_L1(R, 0, e) CW_L1(R, 1, e) CW_L1(R, 2, e) CW_L1(R, 3, e) CW_L1(R, 4, e) CW_L1(R, 5, e) CW_L1(R, 6, e) CW_L1(R, 7, e) }
#define CW_S1(R, b, buf) if (b < NLD) *reinterpret_cast<float4*>(&lds[(buf) * ROWS * LDR + b * 32 * LDR + wrow]) = R##b;
#define CW_LSTORE(buf, R) { CW_S1(R, 0, buf) CW_S1(R, 1, buf) CW_S1(R, 2, buf) CW_S1(R, 3, buf) CW_S1(R, 4, buf) CW_S1(R, 5, buf) CW_S1(R, 6, buf) CW_S1(R, 7, buf) }
#define CW_COMPUTE(buf, e)                                                                            \
    if ((e) < NR && p0 + ((e) % nrp) * PS + slot < p1) { /* wave-uniform: a real round and position */ \
        const float* L = &lds[(buf) * ROWS * LDR];                                                    \
        float4 av[4], bv[4];                                                                          \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                               \
            av[u] = *reinterpret_cast<const float4*>(L + arow + 4 * u);                               \
            bv[u] = *reinterpret_cast<const float4*>(L + brow + 4 * u);                               \
        }                                                                                             \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                               \
            bsum += (bv[u].x + bv[u].y) + (bv[u].z + bv[u].w);                                        \
            acc = mfma32(av[u].x, bv[u].x, acc);                                                      \
            acc = mfma32(av[u].y, bv[u].y, acc);                                                      \
            acc = mfma32(av[u].z, bv[u].z, acc);                                                      \
            acc = mfma32(av[u].w, bv[u].w, acc);                                                      \
        }                                                                                             \
    }
    {
        const int last = NR - 1;  // NR >= 1: every chunk holds at least one position
        CW_GLOAD(0, A)
        CW_GLOAD(min(1, last), B)
        CW_GLOAD(min(2, last), C)
        CW_LSTORE(0, A)
        lds_barrier();
        // every step: request round e+3 into the ring slot just freed, compute round e from LDS, park round
        // e+1 (loaded two steps ago) in the other LDS buffer.  Loads, stores and barriers are unconditional
        // (clamped indices; copies past the last round are redundant) -- only the MFMA block is guarded.
        for (int e = 0; e < NR; e += 3) {
            const int buf = e & 1;  // 3 steps per trip: the parity of the trip's first buffer alternates
            CW_GLOAD(min(e + 3, last), A)
            CW_COMPUTE(buf, e)
            CW_LSTORE(buf ^ 1, B)
            lds_barrier();
            CW_GLOAD(min(e + 4, last), B)
            CW_COMPUTE(buf ^ 1, e + 1)
            CW_LSTORE(buf, C)
            lds_barrier();
            CW_GLOAD(min(e + 5, last), C)
            CW_COMPUTE(buf, e + 2)
            CW_LSTORE(buf ^ 1, A)
            lds_barrier();
        }
    }
#undef CW_L1
#undef CW_S1
#undef CW_COMPUTE
#undef CW_GLOAD
#undef CW_LSTORE
    // position slots > 0 park their tile in LDS (free now); slot 0 adds them in slot order and writes the slab
    float* red = lds;  // [PS - 1][NTP][17][64]
    if (slot > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((slot - 1) * NTP + tp) * 17 + r) * 64 + lane] = acc[r];
        red[(((slot - 1) * NTP + tp) * 17 + 16) * 64 + lane] = bsum;
    }
    __syncthreads();
    if (slot > 0) return;
#pragma unroll
    for (int s2 = 1; s2 < PS; ++s2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(((s2 - 1) * NTP + tp) * 17 + r) * 64 + lane];
        bsum += red[(((s2 - 1) * NTP + tp) * 17 + 16) * 64 + lane];
    }
    float* S = a.slab + ((long)pc * a.K + k) * a.slab_stride;
    const long wrow0 = (long)(kh * a.KWe + kw) * a.CIe + it * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[(wrow0 + mfma_row(r, h)) * a.CO + ot * 32 + bl] = acc[r];
    if (kh == 0 && kw == 0 && it == 0) {
        const long wsize = (long)a.KH * a.KWe * a.CIe * a.CO;
        const float sb = bsum + __shfl_xor(bsum, 32);
        if (h == 0) S[wsize + ot * 32 + bl] = sb;
    }
}
template <int NIT, int NOT>
__global__ __launch_bounds__(256) void k_conv_wgrad(ConvWgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (4 / (NIT * NOT)) * (NIT + NOT) * 32 * 36];
    conv_wgrad_body<NIT, NOT>(a, xcd_contiguous_id(), lds);
}

// slab reduce for all conv layers in ONE launch: grad[k][off + e] = sum_pc slab[pc][k][e]  (fixed order)
// (SlabSeg: dense0_update.h, shared with the Adam role of the Conv_0 weight-gradient launch)
struct SlabReduceArgs {
    SlabSeg seg[3];
    float* grad;
    long gP;  // per-head stride of the gradient arena's small-leaf region (conv leaves sit before Dense_0/kernel)
    int K, n_seg;
};
__global__ __launch_bounds__(256) void k_slab_reduce(SlabReduceArgs a) {
    int si = 0;
#pragma unroll
    for (int i = 1; i < 3; ++i)
        if (i < a.n_seg && (long)blockIdx.x >= a.seg[i].first_block) si = i;
    const SlabSeg& g = a.seg[si];
    const long e = ((long)blockIdx.x - g.first_block) * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (e >= g.wsize + g.bsize) return;
    const float* sp = g.slab + (long)k * g.slab_stride + e;
    const long pstride = (long)a.K * g.slab_stride;
    float s = 0.f;
    int pc = 0;
    for (; pc + 4 <= g.npc; pc += 4) {
        float x0 = sp[(pc + 0) * pstride], x1 = sp[(pc + 1) * pstride], x2 = sp[(pc + 2) * pstride], x3 = sp[(pc + 3) * pstride];
        s = (((s + x0) + x1) + x2) + x3;
    }
    for (; pc < g.npc; ++pc) s += sp[pc * pstride];
    const long o = e < g.wsize ? g.w_off + e : g.b_off + (e - g.wsize);
    a.grad[(long)k * a.gP + o] = s;
}

// --------------------------------------------------------------------------------------------
// Adam over (part of) the arenas (optax.adam, idqn.py:52,106-107) and the step epilogue
// --------------------------------------------------------------------------------------------
// (AdamArgs and the per-thread body adam_thread: dense0_update.h)
__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    warm_kernargs<sizeof(AdamArgs)>();  // (latency-bound launches: the lazily loaded argument fields cost dependent scalar round trips)
    const int k = blockIdx.y;
    if (a.ep_count && blockIdx.x == 0 && threadIdx.x == 0) {
        a.ep_count[k] += 1;
        a.ep_cum[k] = a.ep_cum[k] + (double)a.ep_losses[k];
    }
    adam_thread(a, k, (long)blockIdx.x * 256 + threadIdx.x);
}

// count += 1 (optax count, idqn.py:53), cum_losses += losses in f64 (idqn.py:72)
__global__ void k_step_epilogue(int32_t* count, const float* losses, double* cum, int K, int bump_count) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    if (bump_count) count[k] += 1;
    cum[k] = cum[k] + (double)losses[k];
}

