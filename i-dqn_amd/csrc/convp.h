// Convolutions of the Nature-CNN on the bf16 matrix cores with f32 accuracy ("plane" kernels, gfx950).
//
// Why.  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md: 64 vs 1024 FLOP/clk/SIMD) and
// gfx950 has no xf32.  Every f32 operand is therefore split EXACTLY into three bf16 terms x = x0 + x1 + x2 (8 + 8 + 8
// significand bits, round-to-nearest at every level, each residual is representable) and a product is formed from six
// partial products, down to 2^-23 of it:   a b ~= a2 b0 + a0 b2 + a1 b1 + a1 b0 + a0 b1 + a0 b0   (smallest first),
// accumulated in f32 by v_mfma_f32_32x32x16_bf16: 6 x 32 cycles per 16 k-steps against 8 x 64 for the f32 MFMA (2.7x).
// Conv_0 reads uint8 pixels: u is exact in ONE bf16 plane, the division by 255 (architectures/dqn.py:44) moves into the
// packed kernel (w / 255, one rounding, like x / 255 in the reference), so Conv_0 needs 3 products, not 6.
//
// Layouts (all bf16, "plane" layout):
//   activations  act[slot = net * nb + batch block][hp][wp][plane 0..NP-1][c][32 samples]     row (c) = 64 bytes
//                zero borders materialise SAME padding (flax default, architectures/dqn.py:43-51); NP = 3, Conv_0's
//                input has NP = 1.  16 channels of one plane of one pixel = 1 KiB contiguous = one LDS-DMA instruction.
//   weights      packed once per step per net in MFMA-fragment order:
//                wq[superstep = (kh, 16-channel chunk)][tap kw][32-channel out tile][plane][lane][8]   (1 KiB blocks)
// One layout serves both contraction shapes: the forward / data-gradient kernels sum over channels (k = rows: fragments
// by ds_read_b64_tr_b16, 4 rows x 64 B per half-wave = all 64 banks once), the weight-gradient kernels sum over samples
// (k = the 32 samples of a row: fragments by ds_read_b128, rows XOR-swizzled at LDS-DMA time).
#pragma once
#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#define CP_MAX_STRIPS 4  // input rows one workgroup's output positions may span

// ---- forward / data-gradient launch ---------------------------------------------------------------------------
struct CItem {  // one workgroup: np consecutive output positions (row-major over the variant's OH x OW) of one net
    int net, bb, var, p0, np, pad0, pad1, pad2;
};
struct CVar {  // one sub-convolution of a launch (forward: one; data gradient of a stride-S conv: S * S output parities)
    long w_off;              // bytes from the net's packed weights
    int in_off_h, in_off_w;  // padded input row / column read by output (0, 0), tap (0, 0)
    int OH, OW;
    int out_mul, out_add_h, out_add_w;  // output position (oh, ow) -> (oh * out_mul + out_add_h, ...)
    int pad;
};
struct CFwdArgs {
    const unsigned short* in;   // input planes
    const unsigned short* wq;   // packed weights
    const float* pbase[2];      // f32 parameter arenas: net n < n_first reads pbase[0] + n * pstride, else pbase[1] + (n - n_first) * pstride
    unsigned short* out3;       // output planes or nullptr
    float* out_f32;             // f32 rows [slot][(yh * f32_W + yw) * CO + co][32] or nullptr (Conv_2 -> Dense_0 input)
    const unsigned short* mask3;  // epilogue 1: forward activation planes whose sign masks the result (plane 0 is read)
    float* pb;                  // epilogue 1: per-position sums over the 32 samples [slot][pos][CO] (bias gradient) or nullptr
    long wq_stride;             // bytes per net
    long in_slot, out_slot, mask_slot, f32_slot;  // bytes / bytes / bytes / floats per (net, batch block)
    long b_off, pstride;
    int n_first;
    // work items (no table: a workgroup derives its item from its index): per (net, batch block) slot, variant v has
    // r_cnt[v] balanced ranges of its OH * OW positions, the ranges of variant v start at r_begin[v]
    int items_per_slot, r_begin[4], r_cnt[4];
    int in_split;   // input slot of net n: (in_split > 0 ? n >= in_split : n) * nb + bb
    int range_major;  // 1: consecutive workgroups walk the nets of one position range (nets share the input: Conv_0)
    int nb, n_var, epilogue;
    int KH, NCC, S, SX, CO;           // NCC = 16-channel chunks per tap (1 for Conv_0); SX = pixel chunks per output step
    int pix_bytes, plane_bytes, xstep, row_bytes;  // input geometry in bytes (xstep: between consecutive pixel chunks)
    int out_Wp, out_lo_h, out_lo_w, out_W, out_H;  // out_W x out_H: unpadded grid (pb indexing)
    int mask_Wp, mask_lo_h, mask_lo_w, mask_C;
    int f32_W;
    int row_parts;  // persistent kernel (convp_pp.hip) only: > 0 = every output row is cut into this many items (ranges never cross rows)
    CVar var[4];
};

// ---- weight-gradient launch -----------------------------------------------------------------------------------
struct CWItem {  // one workgroup: positions [p0, p0 + np) of one head, one kernel row (Conv_0: all kernel rows)
    int net, kh, chunk, p0, np, pad0, pad1, pad2;
};
struct CWgradArgs {
    const unsigned short* x;    // forward input planes of the conv
    const unsigned short* dy;   // gradient w.r.t. the conv's output, planes [K][nb][...] zero-bordered
    const float* pb;            // [K * nb][OH * OW][CO] sums of dy over the samples (bias gradient) or nullptr
    // work items (no table: a workgroup derives its item from its index, like the forward kernels): n_chunks balanced
    // chunks of the OH * OW positions; index order (head, chunk, kernel row) or, chunk_major, (chunk, head, kernel row)
    int n_chunks, chunk_major, kh_per_item;  // kh_per_item: kernel rows with an item of their own (Conv_0: 1)
    float* slab;                // [n_chunks][K][slab_stride]: weights [(kh, kw, ci)][co], then bias [co]
    long x_slot, dy_slot, slab_stride;  // bytes, bytes, floats
    int x_shared;               // 1: every head reads slot bb (Conv_0 reads the staged `state`), 0: slot k * nb + bb
    int K, nb, KH, KW, S, CI, CO, OH, OW;
    int x_pix, x_plane, x_row;  // bytes
    int dy_pix, dy_Wp, dy_lo_h, dy_lo_w;
    int PG;                     // positions per LDS stage
    float out_div;              // 1, or 255 for Conv_0 (its input planes hold the raw pixel values)
};

// ---- staging launch: uint8 minibatch -> bf16 plane, and every conv kernel -> packed bf16 planes -------------------
struct PackJob {
    long src_off;   // floats from the net's parameter base (the conv kernel leaf, HWIO)
    long dst_off;   // bytes from the net's packed base
    int n_nets;     // nets this job covers (2K forward, K data gradient)
    int KHv, NQ, NCC, CT;        // virtual conv: kernel rows, taps per superstep, 16-channel chunks, out tiles
    int mode;                    // 0 forward kernel; 1 data-gradient kernel of output parity (rh, rw)
    int KW, CI, CO, S, PLh, PLw, rh, rw, KHs;  // the layer's real geometry (mode 1: KHs = K / S taps per parity)
    int div255;                  // Conv_0: pack w / 255
    long first_block;            // block range of this job inside the pack part of the grid
    int blocks_per_net, pad;
};
struct StageArgs {
    const uint8_t* src[2];  // state, next_state  [B][E] uint8 (nullptr src[1]: one set)
    unsigned short* x1;     // [n_sets][nb][Hp][Wp][C][32] bf16
    long E;
    int B, nb, n_sets, H, W, C, lo_h, lo_w, Hp, Wp;
    int n_prep_blocks;      // blocks [0, n_prep_blocks) stage pixels, the rest pack weights
    const float* const* wbase;
    unsigned short* wq;
    long wq_stride;         // bytes per net
    int n_jobs, K;
    // Adam bias corrections of THIS step, 1 / (1 - b^t) with t = count + 1 (optax, idqn.py:52), written for the K heads by
    // the first pack block: off the critical path of the loss kernel, which did it before (two double pows)
    const int32_t* count;
    float* bcinv;
    float b1, b2;
    // replay-sourced step (idqn_learn_on_replay): the stacked gather happens here.  frames != nullptr: sample b of set
    // (0 state / 1 next_state), channel c = frame ((rows[slot_b][2 set] - (3 - c)) mod n_frames) of the ring, zero where
    // 3 - c >= rows[slot_b][2 set + 1] (frames before the episode start); the scalars of the rows go to act / rew / term
    const uint8_t* frames;
    const int32_t* rows;
    long n_frames, frame_bytes;
    int32_t* act_out;
    float* rew_out;
    uint8_t* term_out;
    PackJob job[8];
};
struct StageSlots { int32_t slot[256]; };  // the sampled element slots of a replay-sourced step, as kernel arguments

// exact three-way bf16 split of two f32 values, round-to-nearest-even at every level (a plain cast: v_cvt_pk_bf16_f32,
// which keeps a NaN a NaN).  Returns the pair packed (v0 in the low half) per plane.
__device__ __forceinline__ void split3_pk(float v0, float v1, unsigned& q0, unsigned& q1, unsigned& q2) {
    bf16x2 h = __builtin_convertvector((f32x2){v0, v1}, bf16x2);
    q0 = __builtin_bit_cast(unsigned, h);
    float r0 = v0 - __uint_as_float(q0 << 16), r1 = v1 - __uint_as_float(q0 & 0xffff0000u);
    h = __builtin_convertvector((f32x2){r0, r1}, bf16x2);
    q1 = __builtin_bit_cast(unsigned, h);
    r0 -= __uint_as_float(q1 << 16);
    r1 -= __uint_as_float(q1 & 0xffff0000u);
    h = __builtin_convertvector((f32x2){r0, r1}, bf16x2);
    q2 = __builtin_bit_cast(unsigned, h);
}

// LDS-DMA from inline asm: 16 B per lane, global (scalar base + per-lane 32-bit byte offset) -> LDS (M0 + 16 * lane).
// hipcc treats its own global_load_lds builtin as an LDS store and drains vmcnt(0) in front of every later LDS read;
// from asm the copy is invisible to that bookkeeping and is ordered by the explicit s_waitcnt vmcnt + s_barrier.
__device__ __forceinline__ void dma16(unsigned voff, unsigned long sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory", "m0");
}

// Touch every 64-byte line of the kernel-argument segment with back-to-back scalar loads at kernel entry.  hipcc loads
// argument fields lazily, next to their first use: a prologue that needs ~60 of them in dependent steps paid 8 scalar
// round trips (measured ~4000 cycles); after this warm-up they all hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
    static_assert(BYTES <= 32 * 64, "kernel arguments longer than 32 cache lines");
    auto kp = (const __attribute__((address_space(4))) u32x4*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int L = (BYTES + 63) / 64;  // lines; all loads of a group are issued before the one wait its asm statement forces
    {
        const u32x4 v0 = kp[0], v1 = kp[L > 1 ? 4 : 0], v2 = kp[L > 2 ? 8 : 0], v3 = kp[L > 3 ? 12 : 0];
        const u32x4 v4 = kp[L > 4 ? 16 : 0], v5 = kp[L > 5 ? 20 : 0], v6 = kp[L > 6 ? 24 : 0], v7 = kp[L > 7 ? 28 : 0];
        asm volatile("" ::"s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(v4), "s"(v5), "s"(v6), "s"(v7));
    }
    if (L > 8) {
        const u32x4 v0 = kp[32], v1 = kp[L > 9 ? 36 : 32], v2 = kp[L > 10 ? 40 : 32], v3 = kp[L > 11 ? 44 : 32];
        const u32x4 v4 = kp[L > 12 ? 48 : 32], v5 = kp[L > 13 ? 52 : 32], v6 = kp[L > 14 ? 56 : 32], v7 = kp[L > 15 ? 60 : 32];
        asm volatile("" ::"s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(v4), "s"(v5), "s"(v6), "s"(v7));
    }
    if (L > 16) {
        const u32x4 v0 = kp[64], v1 = kp[L > 17 ? 68 : 64], v2 = kp[L > 18 ? 72 : 64], v3 = kp[L > 19 ? 76 : 64];
        const u32x4 v4 = kp[L > 20 ? 80 : 64], v5 = kp[L > 21 ? 84 : 64], v6 = kp[L > 22 ? 88 : 64], v7 = kp[L > 23 ? 92 : 64];
        asm volatile("" ::"s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(v4), "s"(v5), "s"(v6), "s"(v7));
    }
    if (L > 24) {
        const u32x4 v0 = kp[96], v1 = kp[L > 25 ? 100 : 96], v2 = kp[L > 26 ? 104 : 96], v3 = kp[L > 27 ? 108 : 96];
        const u32x4 v4 = kp[L > 28 ? 112 : 96], v5 = kp[L > 29 ? 116 : 96], v6 = kp[L > 30 ? 120 : 96], v7 = kp[L > 31 ? 124 : 96];
        asm volatile("" ::"s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(v4), "s"(v5), "s"(v6), "s"(v7));
    }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): waits until at most n of this
// wave's vector-memory operations (LDS-DMA copies included) are outstanding
__device__ __forceinline__ void wait_vmcnt(int n) {
#define CP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        CP_W(1) CP_W(2) CP_W(3) CP_W(4) CP_W(5) CP_W(6) CP_W(7) CP_W(8) CP_W(9) CP_W(10) CP_W(11) CP_W(12) CP_W(13) CP_W(14)
        CP_W(15) CP_W(16) CP_W(17) CP_W(18) CP_W(19) CP_W(20) CP_W(21) CP_W(22) CP_W(23) CP_W(24) CP_W(25) CP_W(26) CP_W(27)
        CP_W(28) CP_W(29) CP_W(30) CP_W(31) CP_W(32)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // 0, or more than the cases cover: wait for all
    }
#undef CP_W
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// host-side launchers (convp_fwd.hip / convp_wgrad.hip / convp_stage.hip)
// stage_bytes: one of the two LDS stage buffers; lds_bytes: whole dynamic LDS (convp_fwd_lds)
int convp_launch_fwd(const CFwdArgs& a, int NPA, int CT, int NQ, int NT, int n_items, size_t stage_bytes, int ring,
                     size_t lds_bytes, hipStream_t q, long long* prof = nullptr);
int convp_fwd_ring(size_t stage_bytes, int NT, int epilogue, size_t budget);  // stage buffers: 3 when they fit the budget
size_t convp_fwd_mask_off(size_t stage_bytes, int NT, int ring, bool planes_out, bool f32_out);
size_t convp_fwd_lds(size_t stage_bytes, int NT, int epilogue, int ring, bool planes_out, bool f32_out);  // prof: per-workgroup phase stamps [n_items][8] (debugging) or nullptr
int convp_launch_wgrad(const CWgradArgs& a, int NPX, int MT, int CT, int n_items, size_t lds_bytes, hipStream_t q);
// a data gradient and a weight gradient side by side in one launch (convp_pair.hip); convp_pair_built: is this pair compiled
bool convp_pair_built(int NPA, int CT, int NQ, int NT, int WNPX, int WCT, int WNTW, int WPG);
int convp_launch_pair(const CFwdArgs& f, int NPA, int CT, int NQ, int NT, int n_f, size_t f_stage, int ring, size_t f_lds,
                      const CWgradArgs& w, int WNPX, int MT, int WCT, int n_w, size_t w_lds, hipStream_t q, long long* prof);
int convp_launch_stage(const StageArgs& a, int n_blocks, hipStream_t q, const StageSlots* slots = nullptr);
int convp_fwd_max_nt(int CT);
// persistent form for launches with several items per CU (convp_pp.hip): n_wg workgroups walk n_items items
bool convp_pp_built(int NPA, int CT, int NQ, int NT);
size_t convp_pp_epi_bytes(int NT, int epilogue, bool planes_out, bool f32_out);
// items: device table of the n_items work items in launch order (row-aligned: pad0 = output row, pad1 = first column)
int convp_launch_fwd_pp(const CFwdArgs& a, int NPA, int CT, int NQ, int NT, int n_items, int n_wg, size_t stage_bytes, int ring,
                        size_t lds_bytes, hipStream_t q, const CItem* items, long long* prof = nullptr);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: the launchers keep one high-water mark per device and
// kernel instantiation (a process that creates handles on a second GPU must set it there too).
struct LdsAttrMark {
    size_t bytes[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // true: the attribute has to be raised to `want` on the current device
    bool needs(size_t want) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return true;  // (unknown device: always set)
        if (want <= bytes[dev]) return false;
        bytes[dev] = want;
        return true;
    }
};

