// f32-accurate convolutions on the bf16 matrix cores ("bf16x3").
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate on gfx950 (MI355X_MICROARCH.md: 64 vs 1024
// FLOP/clk/SIMD) and there is no xf32 form, so the conv layers -- MFMA-bound on the f32 instruction -- use this
// instead: every f32 operand is split EXACTLY into three bf16 terms (x = x0 + x1 + x2: 8 + 8 + 8 significand bits,
// each residual is representable, nothing is lost) and a product is the six partial products down to 2^-24 of it,
//     a b ~= a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0          (dropped: a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b|)
// accumulated in f32 by v_mfma_f32_32x32x16_bf16: 6 x 32 cycles per 16 k-steps against 8 x 64 for the f32 MFMA
// (2.7x), with the same error as an f32 fma chain (tools/probes/bf16x3_probe.hip: 3.0e-7 vs 3.2e-7 of sum |a b|).
//
// Layout.  A "3-plane row" is the batch-minor activation row [(h, w, c)][32 samples] (or a weight row
// [(kh, kw, ci)][32 out channels]) stored as [plane 0 | plane 1 | plane 2] x 32 bf16 = 192 bytes.  With the k index
// (the row) strided and the 32 samples / channels contiguous, MFMA fragments come out of LDS through
// ds_read_b64_tr_b16: a half-wave reads 4 rows x 64 B at a 192-B pitch = banks 0-15 | 48-63 | 32-47 | 16-31,
// conflict-free with no swizzle, and the image is a plain copy of HBM, so LDS-DMA fills it.
#pragma once
#include "cnn_kernels.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef C3_ABLATE
#define C3_ABLATE 0  // debugging only: 1 no DMA in the k-loop, 2 no fragment reads, 4 no MFMAs
#endif
#define ROW3 96  // 16-bit elements per 3-plane row

__device__ __forceinline__ void store3(unsigned short* row, int col, float v) { prep_store3(row, col, v); }

// LDS-DMA issued from inline asm: 16 B per lane, global (per-lane address) -> LDS (wave-uniform byte address in M0
// + 16 * lane).  hipcc treats its own global_load_lds builtin as a store to LDS and puts s_waitcnt vmcnt(0) in front
// of every later ds_read_b64_tr_b16, which serialises the prefetch with the reads it is meant to overlap; from asm
// the copy is invisible to that bookkeeping and is ordered by the explicit s_waitcnt vmcnt(n) + s_barrier below.
__device__ __forceinline__ void glds16u(const unsigned short* gsrc, unsigned short* lds_wave_base) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}

// MFMA operand fragment (32 columns x 16 k) of one plane: lane l -> column l & 31, k = 8 (l >> 5) .. + 7.
// `p` = this lane's transposed-read address: row 8 (l >> 5) + ((l & 15) >> 2), column 16 ((l >> 4) & 1) + 4 (l & 3).
__device__ __forceinline__ bf16x8 frag3(const unsigned short* p) {
    auto q = (const __attribute__((address_space(3))) s16x4*)p;
    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)q);
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + ROW3));  // 4 rows on
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, v0), __builtin_bit_cast(bf16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// --------------------------------------------------------------------------------------------
// weight packing: f32 [KH][KWCI][CO] (a conv kernel in HWIO order, or a transformed data-gradient kernel)
//   -> [KH][KWCI / 32 chunks][CO / 32 tiles][32 rows][3 planes][32 channels] bf16, so that one k-chunk of one
//      32-channel tile is a contiguous 6 KB block (an LDS-DMA copy) whose rows have the 192-B pitch.
// --------------------------------------------------------------------------------------------
struct W3Job {
    long src_off;   // floats from the net's source base
    long dst_off;   // 16-bit elements from the net's packed base
    int src;        // 0: parameter arena (wbase[n]); 1, 2: transformed-weight buffer of Conv_1 / Conv_2
    int n_nets, KH, KWCI, CO;
};
struct W3PackArgs {
    const float* const* wbase;  // [n_nets]
    const float* wt[3];
    long wt_stride[3];
    unsigned short* w3;  // [n_nets][w3_stride]
    long w3_stride;
    W3Job job[8];
};
__global__ __launch_bounds__(256) void k_w3_pack(W3PackArgs a) {
    const W3Job& j = a.job[blockIdx.z];
    const int n = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= j.n_nets || e >= (long)j.KH * j.KWCI * j.CO) return;
    const float* src = j.src == 0 ? a.wbase[n] : a.wt[j.src] + (long)n * a.wt_stride[j.src];
    const float v = src[j.src_off + e];
    const int co = (int)(e % j.CO);
    const long kq = e / j.CO;  // kh * KWCI + q
    const int q = (int)(kq % j.KWCI), kh = (int)(kq / j.KWCI);
    const int JC = j.KWCI / 32, CT = j.CO / 32;
    const long row = (((long)kh * JC + (q >> 5)) * CT + (co >> 5)) * 32 + (q & 31);
    store3(a.w3 + (long)n * a.w3_stride + j.dst_off + row * ROW3, co & 31, v);
}

// --------------------------------------------------------------------------------------------
// convolution / data gradient (same work decomposition as k_conv_fwd: workgroup = all CO channels x NPW output
// positions x 32 samples of one (net, batch block); k-chunk = 32 rows of one kernel row, double-buffered LDS-DMA)
// --------------------------------------------------------------------------------------------
struct Conv3Args {
    const unsigned short* in3;  // [n_in_sets][nb][in_block rows x 3 planes]  zero-bordered
    float* out;                 // f32 result [n_nets][nb][out_block] (nets < f32_nets only) or nullptr
    unsigned short* out3;       // 3-plane result or nullptr
    const float* const* wbase;  // [n_nets] f32 parameter bases (bias)
    const unsigned short* w3;   // [n_nets][w3_stride] packed weights
    int in_split;       // input set of net n: n >= in_split ? 1 : 0 when in_split > 0 (Conv_0), else n itself
    const float* mask;  // epilogue 1: forward activation whose sign masks the result
    long w3_stride, b_off, in_block, out_block, mask_block, n_items;  // blocks in f32 elements (rows x 32)
    int n_nets, nb, npg, n_var, epilogue, f32_nets;
    int KH, KWCI, S, CI, CO, IWp;
    int out_Wp, out_lo_h, out_lo_w, mask_Wp, mask_lo_h, mask_lo_w;
    ConvVariant var[4];  // w_off: 16-bit elements inside the net's packed weights
    long long* prof;     // debug: per-workgroup phase timestamps [n_items][4] or nullptr
};

template <int CT, int RING>  // CT = CO / 32 (1 or 2); RING = LDS buffers (RING - 1 k-chunks in flight)
__global__ __launch_bounds__(256) void k_conv3(Conv3Args a) {
    constexpr int NPW = 4 / CT, NBLK = CT + NPW, BLK = 32 * ROW3, BUF = NBLK * BLK;
    // ONE __shared__ object: [buffer][ CT weight tiles | NPW position blocks ], each 32 rows x 192 B
    __shared__ __attribute__((aligned(16))) unsigned short lds[RING * BUF];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, bl = lane & 31, h = lane >> 5;
    const long long t_start = a.prof ? wall_clock64() : 0;
    int item = xcd_contiguous_id();
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
    const int npos = v.OH * v.OW;
    long xoff[NPW];
#pragma unroll
    for (int p = 0; p < NPW; ++p) {
        int pos = min(pg * NPW + p, npos - 1);
        int oh = pos / v.OW, ow = pos - oh * v.OW;
        xoff[p] = ((long)(oh * a.S + v.in_off_h) * a.IWp + ow * a.S + v.in_off_w) * a.CI * ROW3;
    }
    f32x16 acc;  // the bias joins in the epilogue: its (dependent, cold) loads then overlap the k-loop
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bias[16];
    if (a.epilogue == 0) {
        const float* pbase = a.wbase[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = pbase[a.b_off + ct * 32 + mfma_row(r, h)];
    }
    const int JC = a.KWCI / 32, NC = a.KH * JC;
    const long xrow = (long)a.IWp * a.CI * ROW3;
    // LDS-DMA in 1 KB pieces (one wave instruction = 64 lanes x 16 B).  A chunk is NBLK blocks of 6 pieces; wave w
    // copies pieces w, w + 4, w + 8, ...  Everything that does not change from chunk to chunk is worked out once:
    // the per-lane 32-bit byte offset and the LDS offset of each of the wave's pieces.  In the loop a piece costs a
    // 64-bit scalar base select, s_mov m0 and the load (global_load_lds saddr + voffset form).
    constexpr int NPIECE = NBLK * 6, SLOTS = (NPIECE + 3) / 4;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    unsigned voff[SLOTS], ldo[SLOTS];
    bool from_w[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int j = min(wave_s + 4 * i, NPIECE - 1), blk = j / 6, w6 = j - blk * 6;
        from_w[i] = blk < CT;
        long xo = xoff[0];
#pragma unroll
        for (int p = 1; p < NPW; ++p) xo = (blk - CT == p) ? xoff[p] : xo;
        voff[i] = (unsigned)(w6 * 1024 + lane * 16) + (from_w[i] ? (unsigned)(blk * BLK * 2) : (unsigned)(xo * 2));
        ldo[i] = (unsigned)((blk * BLK + w6 * 512) * 2);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)&lds[0];
    const unsigned long wb0 = (unsigned long)(a.w3 + (long)n * a.w3_stride + v.w_off);
    const unsigned long xb0 = (unsigned long)(a.in3 + ((long)(a.in_split > 0 ? (n >= a.in_split ? 1 : 0) : n) * a.nb + bb) * a.in_block * 3);
    constexpr int PER_MIN = NPIECE / 4;  // DMA instructions per wave and chunk (some waves one more when NBLK is odd)
#define C3_STAGE(c, buf)                                                                              \
    {                                                                                                 \
        const int kh_ = (c) / JC, q0_ = ((c) - kh_ * JC) * 32;                                        \
        const unsigned long sw_ = wb0 + (unsigned long)(c) * (CT * BLK * 2);                          \
        const unsigned long sx_ = xb0 + (unsigned long)(kh_ * xrow + (long)q0_ * ROW3) * 2;           \
        const unsigned lb_ = lds0 + (buf) * (BUF * 2);                                                \
        _Pragma("unroll") for (int i = 0; i < SLOTS; ++i) {                                           \
            if (i < PER_MIN || wave_s + 4 * i < NPIECE)                                               \
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"          \
                             ::"v"(voff[i]), "s"(from_w[i] ? sw_ : sx_), "s"(lb_ + ldo[i]) : "memory", "m0"); \
        }                                                                                             \
    }
#pragma unroll
    for (int c = 0; c < RING - 1; ++c)
        if (c < NC) C3_STAGE(c, c)
    const long long t_loop = a.prof ? wall_clock64() : 0;
    const long long c_loop = a.prof ? clock64() : 0;
    const int lo = (8 * h + ((lane & 15) >> 2)) * ROW3 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    int buf = 0, nbuf = RING - 1;
    for (int c = 0; c < NC; ++c) {
        // chunk c has landed once at most the DMA of the RING - 2 younger chunks is outstanding; the barrier also
        // tells every wave that chunk c - 1's buffer (the one re-filled below) is no longer being read
        if (RING == 2 || c + RING - 2 >= NC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * PER_MIN) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if !(C3_ABLATE & 1)
        if (c + RING - 1 < NC) C3_STAGE(c + RING - 1, nbuf)
#endif
        const unsigned short* as = &lds[buf * BUF + ct * BLK + lo];
        const unsigned short* bs = &lds[buf * BUF + (CT + sub) * BLK + lo];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#if C3_ABLATE & 2
            bf16x8 a0, a1, a2, b0, b1, b2;
            asm volatile("" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(b0), "=v"(b1), "=v"(b2));
#else
            const bf16x8 a0 = frag3(as + ks * 16 * ROW3), a1 = frag3(as + ks * 16 * ROW3 + 32), a2 = frag3(as + ks * 16 * ROW3 + 64);
            const bf16x8 b0 = frag3(bs + ks * 16 * ROW3), b1 = frag3(bs + ks * 16 * ROW3 + 32), b2 = frag3(bs + ks * 16 * ROW3 + 64);
#endif
#if C3_ABLATE & 4
            asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2));
            continue;
#endif
            acc = mfma_bf16(a2, b0, acc);  // smallest terms first
            acc = mfma_bf16(a0, b2, acc);
            acc = mfma_bf16(a1, b1, acc);
            acc = mfma_bf16(a1, b0, acc);
            acc = mfma_bf16(a0, b1, acc);
            acc = mfma_bf16(a0, b0, acc);
        }
        buf = buf + 1 == RING ? 0 : buf + 1;
        nbuf = nbuf + 1 == RING ? 0 : nbuf + 1;
    }
#undef C3_STAGE
    const long long t_epi = a.prof ? wall_clock64() : 0;
    const long long c_epi = a.prof ? clock64() : 0;
    if (a.prof && t == 0) {
        long long* pr = a.prof + (long)blockIdx.x * 4;
        // HW_ID (register 4): cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID (register 20) [3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        pr[0] = t_start; pr[1] = t_loop; pr[2] = t_epi | ((long long)(((xcc & 15) << 8) | ((hw >> 8) & 0xff)) << 48);
    }
    const int pos = pg * NPW + sub;
    if (pos >= npos) return;
    const int oh = pos / v.OW, ow = pos - oh * v.OW;
    const int yh = oh * v.out_mul + v.out_add_h, yw = ow * v.out_mul + v.out_add_w;
    const long row0 = ((long)(yh + a.out_lo_h) * a.out_Wp + (yw + a.out_lo_w)) * (32 * CT) + ct * 32;
    float res[16];
    if (a.epilogue == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = fmaxf(acc[r] + bias[r], 0.f);
    } else {
        const float* M = a.mask + ((long)n * a.nb + bb) * a.mask_block +
                         (((long)(yh + a.mask_lo_h) * a.mask_Wp + (yw + a.mask_lo_w)) * (32 * CT) + ct * 32) * 32 + bl;
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = M[mfma_row(r, h) * 32] > 0.f ? acc[r] : 0.f;  // loads before any store
    }
    if (a.out && n < a.f32_nets) {
        float* Y = a.out + ((long)n * a.nb + bb) * a.out_block;
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[(row0 + mfma_row(r, h)) * 32 + bl] = res[r];
    }
    if (a.out3) {
        unsigned short* Y3 = a.out3 + ((long)n * a.nb + bb) * a.out_block * 3;
#pragma unroll
        for (int r = 0; r < 16; ++r) store3(Y3 + (row0 + mfma_row(r, h)) * ROW3, bl, res[r]);
    }
    if (a.prof && t == 0) a.prof[(long)blockIdx.x * 4 + 3] = c_epi - c_loop;  // shader-clock cycles of the k-loop
}
