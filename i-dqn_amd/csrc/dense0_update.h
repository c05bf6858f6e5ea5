// Dense_0 weight gradient + fused Adam (+ fused data gradient): the workgroup bodies of k_dense0_wgrad / k_dense0_wgrad_pair /
// k_dense0_wgrad_alds (cnn_kernels.h).
// Reference: optax.adam on Dense_0/kernel inside iDQN.learn_on_batch (slimdqn/networks/idqn.py:105-107).
#pragma once
#include <type_traits>

#include "convp.h"

// Cache policy of the Dense_0 streams (tools/probes/mall_policy_probe.hip, profiles/r3_mall_policy_probe.txt): a
// default-policy read that comes behind dirty lines in the memory-side cache pays for their write-back (2.8 instead of
// 6.3 TB/s); a non-temporal read does not allocate, evicts nothing and runs at 5.6-7.0 TB/s in either state.
//   D0_FWD_NT  1: the forward pass streams W (read once per step and net) non-temporally
//   D0_WG_NT   bit 0: theta / m / v loads of the fused update non-temporal; bit 1: its stores
#ifndef D0_FWD_NT
#define D0_FWD_NT 1
#endif
#ifndef D0_WG_NT
#define D0_WG_NT 3
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4(const float* p) {
    if (NT) {
        const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st4(float* p, const float4& v) {
    if (NT) __builtin_nontemporal_store((f32x4v){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4v*>(p));
    else *reinterpret_cast<float4*>(p) = v;
}


// --------------------------------------------------------------------------------------------
// Dense_0 weight gradient (+ optionally fused Adam): g[f][j] = sum_b a3[f][b] * dh[j][b]
// HBM-bound: the output (and, fused, theta/m/v) is the 15.9 MB/head matrix; MFMA work is ~10 % of the time.
// --------------------------------------------------------------------------------------------
struct AdamConsts {
    float lr_neg, b1, b2, omb1, omb2, eps;
};
// The reciprocal bias corrections 1 / (1 - b^t), t = count + 1, are computed ONCE per step and head
// (k_td_dh / k_fc_step, one thread) into bcinv[k][2]; b^t is the correctly rounded f32 power of the
// f32 base (double pow, then rounded).
// optax.adam element update (idqn.py:106-107).  The bias corrections are multiplications by the
// precomputed reciprocals and the final quotient uses the hardware sqrt / rcp (<= 2 ulp each): the
// IEEE-exact division / sqrt expansions made the fused kernel VALU-bound (4.4 k instructions per
// wave) while changing the update by < 1e-6 relative (~1e-11 absolute on a parameter).
__device__ __forceinline__ void adam_elem(const AdamConsts& c, float rbc1, float rbc2, float g, float& th, float& m,
                                          float& v) {
    m = fmaf(c.omb1, g, c.b1 * m);
    v = fmaf(c.omb2, g * g, c.b2 * v);
    const float mh = m * rbc1, vh = v * rbc2;
    const float d = __builtin_amdgcn_sqrtf(vh) + c.eps;
    th = fmaf(c.lr_neg, mh * __builtin_amdgcn_rcpf(d), th);
}

// slab descriptors of the conv weight gradients: grad[k][off + e] = sum over the position chunks pc of slab[pc][k][e]
struct SlabSeg {
    const float* slab;
    long slab_stride, w_off, b_off, wsize;
    int npc, bsize;
    long first_block;  // blockIdx.x range [first_block, first_block + n_blocks)
};

// --------------------------------------------------------------------------------------------
// Adam over (part of) the arenas (optax.adam, idqn.py:52,106-107)
// --------------------------------------------------------------------------------------------
struct AdamArgs {
    float *theta, *mu, *nu;
    const float* grad;
    const float* bcinv;  // [K][2]
    AdamConsts ad;
    long P, begin, end;         // element range inside a head, multiples of 4
    long skip_begin, skip_end;  // sub-range already updated by a fused kernel (empty when begin==end)
    long gP, w0_begin, w0_end, g_w0_base;  // gradient arena layout (GradLayout in qnet.hip)
    int K, n_seg;               // n_seg > 0: the conv leaves' gradients are still per-chunk slabs (fused path):
    SlabSeg seg[3];             // sum them here (fixed chunk order) instead of a separate reduce launch
    // step epilogue of the two-phase (data-parallel) step, or nullptr: count += 1 (optax count, idqn.py:53) and
    // cum_losses += losses in f64 (idqn.py:72), by one thread per head -- this launch reads bcinv, not count
    int32_t* ep_count;
    const float* ep_losses;
    double* ep_cum;
};
// One thread of the small-leaf Adam launch (k_adam, cnn_kernels.h; also the ADAM ROLE of the Conv_0 weight-gradient launch,
// convp_wgrad.hip): thread gid of head k.
__device__ __forceinline__ void adam_thread(const AdamArgs& a, const int k, const long gid) {
    // FOUR lanes per float4 of parameters: where the gradient is still per-chunk slabs (the conv leaves on the fused path),
    // each lane sums every fourth chunk, up to 8 loads in flight, and the four partial sums are combined in a
    // fixed order, ((l0 + l1) + (l2 + l3)) -- one latency round instead of npc / 8 (Conv_0 has 51 chunks).  Lane 0 of the
    // quad then does the update.  The grid covers [begin, end) minus the skipped range (the fused kernel's
    // Dense_0/kernel: 98 % of the head), compacted.
    const int sub = (int)(gid & 3);
    long e = a.begin + (gid >> 2) * 4;
    if (e >= a.skip_begin) e += a.skip_end - a.skip_begin;
    if (e >= a.end) return;  // whole quads leave together
    const float bc1 = a.bcinv[2 * k], bc2 = a.bcinv[2 * k + 1];
    const long o = (long)k * a.P + e;
    const long w0n = a.w0_end - a.w0_begin;
    const long go = e < a.w0_begin ? (long)k * a.gP + e
                    : (e < a.w0_end ? a.g_w0_base + (long)k * w0n + (e - a.w0_begin) : (long)k * a.gP + e - w0n);
    float4 th = make_float4(0.f, 0.f, 0.f, 0.f), m = th, v = th, g = th;
    if (sub == 0) {  // independent of the gradient assembly below: issue first
        th = *reinterpret_cast<float4*>(a.theta + o);
        m = *reinterpret_cast<float4*>(a.mu + o);
        v = *reinterpret_cast<float4*>(a.nu + o);
    }
    bool from_slab = false;
#pragma unroll
    for (int si = 0; si < 3; ++si) {
        if (si >= a.n_seg) break;
        const SlabSeg& sg = a.seg[si];
        long se = -1;  // element index inside the slab (weights, then bias); leaves are 4-aligned
        if (e >= sg.w_off && e < sg.w_off + sg.wsize) se = e - sg.w_off;
        else if (e >= sg.b_off && e < sg.b_off + sg.bsize) se = sg.wsize + (e - sg.b_off);
        if (se >= 0) {
            from_slab = true;
            const float* sp = sg.slab + (long)k * sg.slab_stride + se;
            const long pstride = (long)a.K * sg.slab_stride;
            for (int pc0 = sub; pc0 < sg.npc; pc0 += 32) {  // 8 chunks per lane and round (32 per quad), no load past the last chunk
                float4 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (pc0 + 4 * u < sg.npc) x[u] = ld4<true>(sp + (pc0 + 4 * u) * pstride);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (pc0 + 4 * u < sg.npc) { g.x += x[u].x; g.y += x[u].y; g.z += x[u].z; g.w += x[u].w; }
            }
        }
    }
    if (from_slab) {  // quad-uniform: the four lanes share e
        g.x += __shfl_xor(g.x, 1); g.y += __shfl_xor(g.y, 1); g.z += __shfl_xor(g.z, 1); g.w += __shfl_xor(g.w, 1);
        g.x += __shfl_xor(g.x, 2); g.y += __shfl_xor(g.y, 2); g.z += __shfl_xor(g.z, 2); g.w += __shfl_xor(g.w, 2);
    } else if (sub == 0) {
        g = *reinterpret_cast<const float4*>(a.grad + go);
    }
    if (sub != 0) return;
    adam_elem(a.ad, bc1, bc2, g.x, th.x, m.x, v.x);
    adam_elem(a.ad, bc1, bc2, g.y, th.y, m.y, v.y);
    adam_elem(a.ad, bc1, bc2, g.z, th.z, m.z, v.z);
    adam_elem(a.ad, bc1, bc2, g.w, th.w, m.w, v.w);
    *reinterpret_cast<float4*>(a.theta + o) = th;
    *reinterpret_cast<float4*>(a.mu + o) = m;
    *reinterpret_cast<float4*>(a.nu + o) = v;
}

struct DenseWgradArgs {
    const float* a3;  // [2K][nb][F*32]
    const float* dh;  // [K][nb][J][32]
    float* grad;      // [K][P]
    float *theta, *mu, *nu;
    const float* bcinv;  // [K][2]
    AdamConsts ad;
    long w_off, P, n_items;
    long g_w0_base, g_w0_stride;  // unfused output: grad + g_w0_base + k * g_w0_stride (contiguous over heads)
    // sample block bb of head k: base + (bb / nb_inner) * outer + k * head + (bb % nb_inner) * inner   (floats)
    long a3_outer, a3_head, a3_inner, dh_outer, dh_head, dh_inner;
    int K, nb, nb_inner, n_ft, n_jt, F, J;
    int keep_heads;  // host side only: heads [0, keep_heads) are launched with the kernel that stores theta_new with the default cache policy
    int item0;     // stand-alone kernel: workgroup b takes item b + item0 (the tail of an update that conv launches began)
    // NQ = 4 (full 512-column rows) with FUSE_DG: the workgroup finishes dL/da3 itself -- what k_da3_finalize does otherwise
    unsigned short* da3p;  // bf16 planes (convp.h layout, geometry g) or nullptr
    float* da3f;           // f32 rows [K][nb][g.block] (f32 conv path) or nullptr
    float* pb;             // [K * nb][H * W][C] sums over the 32 samples (Conv_2 bias gradient) or nullptr
    ActGeom g;
    int C;
    float* dpart;  // FUSE_DG: partial data gradients [n_jt][K * nb][F][32] (this column tile's share of dL/da3), else unused
    // BF3: the two factors once more as three exact bf16 planes (k_split_factors), compact: a3p[plane][bb][k][F * 32],
    // dhp[plane][bb][k][J * 32]
    const unsigned short *a3p, *dhp;
};


// Workgroup = one 32 (f) x 256 (j) tile of one head.  Phase 1: each of the 4 waves computes a 32 x 64 sub-tile
// on the MFMA (2 accumulators, k = the 32 samples per batch block) and parks it in LDS.  Phase 2: all 256
// threads stream the tile row by row -- every wave-instruction moves one whole 1 KB row segment of theta / m / v
// (16 B per lane), the gradient comes from LDS -- exactly the access pattern of the plain Adam kernel, which
// reaches 6.9 TB/s on MI355X.  The first version kept the tile in accumulators and ran Adam from the MFMA
// layout at 2 waves/SIMD (5.3 TB/s); this one needs ~100 registers and 32 KB of LDS (4-5 workgroups per CU).
#ifndef D0W_DEPTH_ROWS
#define D0W_DEPTH_ROWS 8
#endif
#ifndef D0W_DEPTH
#define D0W_DEPTH 4  // row groups of theta / m / v in flight per thread in the fused kernel's streaming phase
#endif
// FUSE_DG (with FUSE_ADAM): the workgroup also produces its column tile's share of the Dense_0 DATA gradient,
//   dL/da3[f][b] += sum over its 256 columns j of theta_old[f][j] * dh[j][b],
// from the theta rows that stream through its registers anyway -- the separate data-gradient kernel re-read all of
// theta (79 MB per step at K = 5; a pure read stream runs at ~4 TB/s on this chip: 24 us).  Each thread parks the
// pre-update theta float4 in the LDS slot whose gradient it has just consumed; after the streaming phase the four waves
// run D'[b][f] over 64 columns each on the MFMA (B operand = theta from LDS, A operand = dh rows from L2), add their four
// tiles in LDS and write the 4 KB partial.  k_da3_finalize sums the column tiles' partials and applies the ReLU mask.
// The LDS tile's columns are rotated by 4 * row: the MFMA reads one column of 32 rows per instruction, which would hit a
// single bank with a 256-float pitch (rotated: 4-way, 8 cycles per 64-cycle MFMA); float4 accesses stay aligned.
// RT = 2 (with NQ = 1, BF3): the tile is 64 rows x 128 columns instead of 32 x 256 -- the same 8192 elements, the same four
// 32 x 64 wave tiles (wave w: row half w >> 1, column half w & 1), the same number of workgroups, but (64 + 128) instead of
// (32 + 256) operand rows per sample block: a third less L2 -> CU operand traffic, which is what the N-block contraction of
// the factored data-parallel update is bound by (qnet.hip, launch_dense0_wgrad); theta / m / v stream as 512-byte row pieces.
// ALDS (with BF3, RT = 1, NQ = 2): the contraction over the N sample blocks of the factored data-parallel update was one
// DEPENDENT round trip per block -- a tile's a3 planes are read exactly twice (once per column tile), i.e. they come from HBM,
// and the 18 fragments of a block leave no registers for a second block in flight (7.7 us per extra block).  Here the tile's a3
// fragments of up to 8 blocks (6 KB each) are copied up front by LDS-DMA, in fragment order, into the LDS the gradient tile is
// parked in afterwards (+ 16 KB): ONE HBM round trip per tile; the dh fragments (L2-resident: 0.8 MB per head) stay in registers,
// two blocks ahead.  RT = 2 with ALDS: a 64 x 256 tile -- the two row tiles share every dh fragment (half the L2 -> CU operand
// traffic, twice the products behind every fragment wait), 64 accumulator registers, two workgroups per CU.
template <bool FUSE_ADAM, int NQ, bool FUSE_DG, bool BF3, int RT = 1, bool ALDS = false, bool TH_ST_NT = (D0_WG_NT & 2) != 0>  // column tile JT = 128 * NQ (256 when the dense width allows it); TH_ST_NT: see dense0_pair_body
__device__ __forceinline__ void dense0_wgrad_body(const DenseWgradArgs& a, int item, float* gs /* LDS, 32 * RT * JT floats (+ 4096 FUSE_DG) */,
                                                  const int t /* 0..255: thread of the 256-thread group that owns the item */) {
    constexpr int JT = 128 * NQ, LPR = JT / 4, RPI = 1024 / JT, NIT = 32 * RT / RPI;  // lanes/row, rows/iter (256 threads), iters
    static_assert(RT == 1 || (RT == 2 && (NQ == 1 || ALDS) && BF3 && !FUSE_DG), "64-row tiles: the bf16-plane update without the fused data gradient");
    constexpr bool TALL = ALDS && RT == 2;  // 64 x 256 tile: every wave keeps 64 columns of BOTH row halves (two tiles share the dh fragments)
    constexpr int NQW = TALL ? 4 : RT == 1 ? NQ : 2;  // 32 x 32 accumulator tiles per wave
    static_assert(!FUSE_DG || (FUSE_ADAM && (NQ == 2 || NQ == 4)), "the fused data gradient rides on the fused 256- / 512-column kernels");
    static_assert(!ALDS || (BF3 && NQ == 2), "a3 fragments through LDS: the 32 x 256 / 64 x 256 bf16-plane update");
    constexpr bool ROWS = FUSE_DG && NQ == 4;  // whole rows: the data gradient is complete here
    const int lane = t & 63, wave = t >> 6, bl = lane & 31, h = lane >> 5;
#if D0W_JT_SLOW  // locality experiment: the two column tiles of a row range far apart in dispatch order (jt slowest)
    const int ft = item % a.n_ft;
    item /= a.n_ft;
    const int k = item % a.K;
    const int jt = item / a.K;
#else
    const int jt = item % a.n_jt;
    item /= a.n_jt;
    const int ft = item % a.n_ft;
    const int k = item / a.n_ft;
#endif
    const int f0 = ft * 32 * RT, j0 = jt * JT, jw = (RT == 1 || TALL) ? wave * (32 * NQ) : (wave & 1) * 64,
              rw = (RT == 1 || TALL) ? 0 : (wave >> 1) * 32;
    const long base = (long)k * a.P + a.w_off + (long)f0 * a.J + j0;
    // phase-2 addressing: iteration i, this thread: row RPI * i + prow, columns pcol .. pcol + 3
    const int prow = t / LPR, pcol = (t % LPR) * 4;
    const long o0 = base + (long)prow * a.J + pcol;
    auto rot = [&](int row, int col) { return FUSE_DG ? row * JT + ((col + 4 * row) & (JT - 1)) : row * JT + col; };
    // phase-2 state runs DEPTH row groups ahead (a ring of named-index registers): the rows of the first DEPTH
    // iterations are requested before the MFMA phase, so the workgroup keeps streaming while it computes its tile
    // (the full-row kernel has two workgroups per CU instead of three: deeper, to keep as many bytes in flight per CU)
    constexpr int DEPTH = (FUSE_DG && NQ == 4) ? D0W_DEPTH_ROWS : D0W_DEPTH;
    float4 th[DEPTH], mm[DEPTH], vv[DEPTH];
    auto prefetch = [&]() {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long on = o0 + (long)(RPI * d) * a.J;
            th[d] = ld4<(D0_WG_NT & 1) != 0>(a.theta + on);
            mm[d] = ld4<(D0_WG_NT & 1) != 0>(a.mu + on);
            vv[d] = ld4<(D0_WG_NT & 1) != 0>(a.nu + on);
        }
    };
#ifndef D0W_JT_SLOW
#define D0W_JT_SLOW 0
#endif
#ifndef D0W_ALDS_PRE
#define D0W_ALDS_PRE 0  // 0: the ALDS variant requests its first row groups only after the contraction (48 registers less in the block loop)
#endif
#ifndef D0W_PRE_OPS
#define D0W_PRE_OPS 1  // the f32 contraction's operand loads go out in FRONT of the first row groups' (an in-order wait for the
#endif                 // operands then does not include the twelve HBM loads of the prefetch)
    constexpr bool PRE_OPS = D0W_PRE_OPS && FUSE_ADAM && !BF3 && !ALDS && NQ <= 2;  // (the full-row kernel has no registers for it)
    constexpr bool PRE_EARLY = (!ALDS || D0W_ALDS_PRE) && !PRE_OPS;
    if (FUSE_ADAM && PRE_EARLY) prefetch();
    f32x16 acc[NQW];
#pragma unroll
    for (int q = 0; q < NQW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    if constexpr (ALDS) {
        constexpr int CH = 8 / RT;  // sample blocks staged at once: CH * RT row tiles * 3 planes * 2 KB = 48 KB
        const long pa = (long)a.nb * a.K * a.F * 32, pd = (long)a.nb * a.K * a.J * 32;
        const unsigned lds0 = (unsigned)(uintptr_t)gs;  // low half of a generic LDS address = the LDS byte address
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        // fragment order: piece ((block * RT + row tile) * 3 + plane, step s) is 1 KB = lane (h, bl)'s 16 bytes: row f0 + 32 rt + bl,
        // samples 16 h + 8 s ..
        const unsigned voff = (unsigned)(bl * 64 + 32 * h);
        // dh fragments: wave-uniform base (scalar registers) + the lane's 32-bit byte offset, the same one as the a3 copies
        const unsigned short* Dp0 = a.dhp + (long)(j0 + wv * 64) * 32;
        auto load_b = [&](int bb, int q, bf16x8 (&Bq)[3][2]) {
            const unsigned char* Dp = reinterpret_cast<const unsigned char*>(Dp0 + ((long)bb * a.K + k) * a.J * 32 + (long)q * 32 * 32);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) Bq[pl][s2] = *reinterpret_cast<const bf16x8*>(Dp + (pl * pd + 8 * s2) * 2 + voff);
        };
        for (int c0 = 0; c0 < a.nb; c0 += CH) {
            const int nc = a.nb - c0 < CH ? a.nb - c0 : CH;
            if (c0 > 0) __syncthreads();  // every wave has read the previous chunk's fragments
            if (RT == 1 && a.a3p == nullptr) {
                // The tile's f32 rows of the chunk's blocks straight from the (gathered) factors: 16 bytes per thread and block, all
                // requested at once (one round trip, as the copies below), split here and stored in the same fragment order -- no
                // plane copy of a3 in HBM (k_split_factors then only splits dh: 42 MB read + 63 MB written less per step at 8 blocks).
                // Every a3 tile is split by the two column-tile workgroups that read it; the same split3_pk: bit-identical.
                f32x4v xv[CH];
                const int row = t >> 3, j8 = t & 7;
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int bb = c0 + (u < nc ? u : 0), bo = bb / a.nb_inner, bi = bb - bo * a.nb_inner;
                    xv[u] = *reinterpret_cast<const f32x4v*>(a.a3 + bo * a.a3_outer + k * a.a3_head + bi * a.a3_inner + (long)(f0 + row) * 32 + 4 * j8);
                }
                unsigned char* dst0 = reinterpret_cast<unsigned char*>(gs) + ((j8 >> 1) & 1) * 1024 + ((j8 >> 2) * 32 + row) * 16 + 8 * (j8 & 1);
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (u < nc) {
                        unsigned pa0, pa1, pa2, pb0, pb1, pb2;
                        split3_pk(xv[u].x, xv[u].y, pa0, pa1, pa2);
                        split3_pk(xv[u].z, xv[u].w, pb0, pb1, pb2);
                        unsigned char* d = dst0 + u * 6 * 1024;
                        *reinterpret_cast<u32x2*>(d) = (u32x2){pa0, pb0};
                        *reinterpret_cast<u32x2*>(d + 2048) = (u32x2){pa1, pb1};
                        *reinterpret_cast<u32x2*>(d + 4096) = (u32x2){pa2, pb2};
                    }
                }
            } else
            for (int pc = wv; pc < nc * 6 * RT; pc += 4) {  // 1 KB pieces, dealt to the four waves
                const int c = pc >> 1, s2 = pc & 1, br = c / 3, pl = c - 3 * br, bbl = br / RT, rt = br - RT * bbl;
                const unsigned short* src = a.a3p + pl * pa + ((long)(c0 + bbl) * a.K + k) * a.F * 32 + (long)(f0 + 32 * rt) * 32 + 8 * s2;
                dma16(voff, (unsigned long)src, lds0 + (unsigned)pc * 1024);
            }
            bf16x8 B[4][3][2];  // ring of four (block parity, q) units: a unit is re-filled two blocks ahead once its products are issued
            load_b(c0, 0, B[0]);
            load_b(c0, 1, B[1]);
            if (nc > 1) {
                load_b(c0 + 1, 0, B[2]);
                load_b(c0 + 1, 1, B[3]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned char* lA = reinterpret_cast<const unsigned char*>(gs) + lane * 16;
#pragma unroll 1
            for (int bb2 = 0; bb2 < nc; bb2 += 2) {
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const int bbl = bb2 + par;
                    if (bbl < nc) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {  // per accumulator the product order of the register version
#pragma unroll
                                for (int rt = 0; rt < RT; ++rt) {
                                    bf16x8 A[3];
#pragma unroll
                                    for (int pl = 0; pl < 3; ++pl)
                                        A[pl] = *reinterpret_cast<const bf16x8*>(lA + (((bbl * RT + rt) * 3 + pl) * 2 + s2) * 1024);
                                    f32x16& d = acc[2 * rt + q];
                                    d = mfma_bf16(A[2], B[2 * par + q][0][s2], d);
                                    d = mfma_bf16(A[0], B[2 * par + q][2][s2], d);
                                    d = mfma_bf16(A[1], B[2 * par + q][1][s2], d);
                                    d = mfma_bf16(A[1], B[2 * par + q][0][s2], d);
                                    d = mfma_bf16(A[0], B[2 * par + q][1][s2], d);
                                    d = mfma_bf16(A[0], B[2 * par + q][0][s2], d);
                                }
                            }
                            if (bbl + 2 < nc) load_b(c0 + bbl + 2, q, B[2 * par + q]);
                        }
                    }
                }
            }
        }
        if (!PRE_EARLY) prefetch();
        __syncthreads();  // the gradient tile is parked over the fragments
    } else
    for (int bb = 0; bb < a.nb; ++bb) {
        if (BF3) {
            // lane (bl, h): MFMA step s, element i = sample 16 h + 8 s + i for both operands; six products, smallest first
            const long slot = (long)bb * a.K + k, pa = (long)a.nb * a.K * a.F * 32, pd = (long)a.nb * a.K * a.J * 32;
            const unsigned short* Ap = a.a3p + slot * a.F * 32 + (long)(f0 + rw + bl) * 32 + 16 * h;
            const unsigned short* Dp = a.dhp + slot * a.J * 32 + (long)(j0 + jw + bl) * 32 + 16 * h;
            bf16x8 A[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int s = 0; s < 2; ++s) A[pl][s] = *reinterpret_cast<const bf16x8*>(Ap + pl * pa + 8 * s);
#pragma unroll
            for (int q = 0; q < NQW; ++q) {
                bf16x8 B[3][2];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int s = 0; s < 2; ++s) B[pl][s] = *reinterpret_cast<const bf16x8*>(Dp + (long)q * 32 * 32 + pl * pd + 8 * s);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc[q] = mfma_bf16(A[2][s], B[0][s], acc[q]);
                    acc[q] = mfma_bf16(A[0][s], B[2][s], acc[q]);
                    acc[q] = mfma_bf16(A[1][s], B[1][s], acc[q]);
                    acc[q] = mfma_bf16(A[1][s], B[0][s], acc[q]);
                    acc[q] = mfma_bf16(A[0][s], B[1][s], acc[q]);
                    acc[q] = mfma_bf16(A[0][s], B[0][s], acc[q]);
                }
            }
            continue;
        }
        const int bo = bb / a.nb_inner, bi = bb - bo * a.nb_inner;
        const float* Ap = a.a3 + bo * a.a3_outer + k * a.a3_head + bi * a.a3_inner + (long)(f0 + bl) * 32 + 16 * h;
        float4 x0 = *reinterpret_cast<const float4*>(Ap), x1 = *reinterpret_cast<const float4*>(Ap + 4);
        float4 x2 = *reinterpret_cast<const float4*>(Ap + 8), x3 = *reinterpret_cast<const float4*>(Ap + 12);
        const float av[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w,
                              x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
        const float* Dp = a.dh + bo * a.dh_outer + k * a.dh_head + bi * a.dh_inner + (long)(j0 + jw + bl) * 32 + 16 * h;
        float4 y[NQ][4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float* dq = Dp + (long)q * 32 * 32;  // columns jw + 32 q + bl
#pragma unroll
            for (int u = 0; u < 4; ++u) y[q][u] = *reinterpret_cast<const float4*>(dq + 4 * u);
        }
        if (PRE_OPS && bb == 0) {
            __builtin_amdgcn_sched_barrier(0);
            prefetch();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float bv[16] = {y[q][0].x, y[q][0].y, y[q][0].z, y[q][0].w, y[q][1].x, y[q][1].y, y[q][1].z, y[q][1].w,
                                  y[q][2].x, y[q][2].y, y[q][2].z, y[q][2].w, y[q][3].x, y[q][3].y, y[q][3].z, y[q][3].w};
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[q] = mfma32(av[u], bv[u], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) gs[rot(rw + (TALL ? 32 * (q >> 1) : 0) + mfma_row(r, h), jw + 32 * (TALL ? q & 1 : q) + bl)] = acc[q][r];
    __syncthreads();
    // phase 3's dh values of the first sample block (L2-resident) are requested HERE, in front of the streaming phase: behind it
    // they queued after the tile's last loads and stores and every workgroup opened phase 3 with a full round trip (32 registers,
    // the kernel stays at three waves per SIMD)
    constexpr int NCH3 = FUSE_DG ? 2 * NQ : 1;  // chunks of 8 MFMA steps (16 columns each) over this wave's 32 * NQ columns
#ifndef D0W_PF3
#define D0W_PF3 2  // chunks requested in front of the streaming phase; the others at the start of phase 3, behind them
#endif
#ifndef D0W_PF3_MID
#define D0W_PF3_MID 1
#endif
    constexpr int PF3 = (FUSE_DG && NQ == 2) ? D0W_PF3 : 0;
    float dv[PF3 > 0 ? NCH3 : 1][8];
    auto load_dv = [&](int bb, auto c_lo, auto c_hi) {
        const int bo = bb / a.nb_inner, bi = bb - bo * a.nb_inner;
        // A operand: dh^T, lane (b = bl, k = h) reads dh[j0 + jw + 2 t + h][b]
        const float* Dp = a.dh + bo * a.dh_outer + k * a.dh_head + bi * a.dh_inner + (long)(j0 + jw + h) * 32 + bl;
#pragma unroll
        for (int c = c_lo; c < c_hi; ++c)
#pragma unroll
            for (int u = 0; u < 8; ++u) dv[c][u] = Dp[(long)(16 * c + 2 * u) * 32];
    };
    if constexpr (PF3 > 0) load_dv(0, std::integral_constant<int, 0>{}, std::integral_constant<int, PF3>{});
    if (FUSE_ADAM) {
        const float bc1 = a.bcinv[2 * k], bc2 = a.bcinv[2 * k + 1];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int s = i % DEPTH;
            float* gp = &gs[rot(RPI * i + prow, pcol)];
            const float4 g = *reinterpret_cast<const float4*>(gp);
            float4 t4 = th[s], m4 = mm[s], v4 = vv[s];
            if constexpr (PF3 > 0 && D0W_PF3_MID) {  // the other chunks go out in the last iterations, into the registers of ring slots
                if (i >= NIT - 3 && PF3 + i - (NIT - 3) < NCH3 && i < NIT - 1) {  // that are not re-filled any more: phase 3 then
                    const int c3 = PF3 + i - (NIT - 3);                            // waits for no store of the last iterations
                    const float* Dp = a.dh + k * a.dh_head + (long)(j0 + jw + h) * 32 + bl;
#pragma unroll
                    for (int u = 0; u < 8; ++u) dv[c3][u] = Dp[(long)(16 * c3 + 2 * u) * 32];
                }
            }
            if (FUSE_DG) *reinterpret_cast<float4*>(gp) = t4;  // theta BEFORE the update takes the consumed gradient's place
            if (i + DEPTH < NIT) {  // the slot just read is re-filled DEPTH row groups ahead, before the (may-alias) stores
                const long on = o0 + (long)(RPI * (i + DEPTH)) * a.J;
                th[s] = ld4<(D0_WG_NT & 1) != 0>(a.theta + on);
                mm[s] = ld4<(D0_WG_NT & 1) != 0>(a.mu + on);
                vv[s] = ld4<(D0_WG_NT & 1) != 0>(a.nu + on);
            }
            adam_elem(a.ad, bc1, bc2, g.x, t4.x, m4.x, v4.x);
            adam_elem(a.ad, bc1, bc2, g.y, t4.y, m4.y, v4.y);
            adam_elem(a.ad, bc1, bc2, g.z, t4.z, m4.z, v4.z);
            adam_elem(a.ad, bc1, bc2, g.w, t4.w, m4.w, v4.w);
            const long o = o0 + (long)(RPI * i) * a.J;
            st4<TH_ST_NT>(a.theta + o, t4);
            st4<(D0_WG_NT & 2) != 0>(a.mu + o, m4);
            st4<(D0_WG_NT & 2) != 0>(a.nu + o, v4);
        }
    } else {
        const long g0 = a.g_w0_base + (long)k * a.g_w0_stride + (long)(f0 + prow) * a.J + j0 + pcol;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            *reinterpret_cast<float4*>(a.grad + g0 + (long)(RPI * i) * a.J) =
                *reinterpret_cast<const float4*>(&gs[(RPI * i + prow) * JT + pcol]);
    }
    if (FUSE_DG) {
        __syncthreads();  // the LDS tile now holds theta_old[32][256] (rotated)
        float* red = gs + 32 * JT;  // [4 waves][32 f][32 b], 16-byte slots XOR-swizzled by (f & 7)
        for (int bb = 0; bb < a.nb; ++bb) {
            // A operand: dh^T (dv); B operand: theta_old[f = bl][j = jw + 2 t + h]
            [[maybe_unused]] const int bo = bb / a.nb_inner, bi = bb - bo * a.nb_inner;
            if constexpr (PF3 > 0) {
                if (bb > 0) load_dv(bb, std::integral_constant<int, 0>{}, std::integral_constant<int, NCH3>{});
                else if (!(D0W_PF3_MID && NCH3 - PF3 <= 2)) load_dv(0, std::integral_constant<int, PF3>{}, std::integral_constant<int, NCH3>{});
            }
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            if constexpr (PF3 > 0) {
#pragma unroll
                for (int c = 0; c < NCH3; ++c) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int col = jw + 16 * c + 2 * u + h;
                        d = mfma32(dv[c][u], gs[bl * JT + ((col + 4 * bl) & (JT - 1))], d);
                    }
                }
            } else {  // nothing requested ahead (the full-row kernel, the last-arriver variant): chunk c + 1 under chunk c's products
                const float* Dp = a.dh + bo * a.dh_outer + k * a.dh_head + bi * a.dh_inner + (long)(j0 + jw + h) * 32 + bl;
                float db[2][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) db[0][u] = Dp[(long)(2 * u) * 32];
#pragma unroll
                for (int c = 0; c < NCH3; ++c) {
                    if (c + 1 < NCH3) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) db[(c + 1) & 1][u] = Dp[(long)(16 * (c + 1) + 2 * u) * 32];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int col = jw + 16 * c + 2 * u + h;
                        d = mfma32(db[c & 1][u], gs[bl * JT + ((col + 4 * bl) & (JT - 1))], d);
                    }
                }
            }
            // this wave's tile -> LDS: lane = row f (bl), 4 x 4 consecutive samples per register quad
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&red[wave * 1024 + bl * 32 + (((2 * g + h) ^ (bl & 7)) * 4)]) =
                    make_float4(d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]);
            __syncthreads();
            {   // all 256 threads: float4 t of the 32 x 32 tile = row t >> 3, slot t & 7; the four waves' tiles in wave order
                const int row = t >> 3, slot = ((t & 7) ^ (row & 7)) * 4;
                float4 s4 = *reinterpret_cast<const float4*>(&red[row * 32 + slot]);
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float4 y = *reinterpret_cast<const float4*>(&red[w * 1024 + row * 32 + slot]);
                    s4.x += y.x; s4.y += y.y; s4.z += y.z; s4.w += y.w;
                }
                float* O = a.dpart + (((long)jt * a.K + k) * a.nb + bb) * a.F * 32 + (long)f0 * 32;  // finished by k_da3_finalize
                *reinterpret_cast<float4*>(O + t * 4) = s4;
            }
            __syncthreads();  // red is reused by the next batch block
        }
    }
}

// ---- the fused update on PAIRS of column tiles (round 4; one sample block, J = 512) ----------------------------------------
// One workgroup takes BOTH 32 x 256 column tiles of its 32 rows, one after the other.  Per tile the three phases of
// dense0_wgrad_body (f32 contraction -> gradient tile in LDS; streaming Adam that leaves theta_old in LDS; data-gradient
// products from LDS), in the same arithmetic and the same order, so every parameter and every partial sum is bit-identical.
// What the pairing buys (the ablations of DESIGN 3.5c put 9 us of this kernel on workgroups that are not streaming):
//  * the second tile's contraction operands are requested before the first tile's phase 3 and its first four row groups right
//    after that phase's products, into registers the first tile has released -- its phase 1 starts with the operands there
//    and its stream never runs dry across the two MFMA phases in between;
//  * the workgroup holds both column tiles' partial data gradients: it adds them (tile 0 + tile 1, the order k_da3_finalize
//    uses), applies the ReLU mask and writes the output forms itself.  No partials in HBM, no finalize launch, and -- unlike the
//    last-arriver variant -- no hand-off either.
// Same registers (vmcnt is in order: operands in front of prefetches, phase-3 operands in front of / inside the stream).
// ROWPAIR: the pair is two ROW tiles (f0, f0 + 32) of ONE column tile instead -- the sibling workgroup (other column tile, adjacent
// in dispatch order) streams the other halves of the same rows at the same time, as in the tile kernel; the partial data
// gradients go to HBM and k_da3_finalize runs as before.
// DEPTH = 8 (= every row group of a tile; two waves per SIMD): the whole tile is requested at once, and every ring slot the first
// tile's stream has consumed is re-filled with the SECOND tile's row group at once -- the workgroup's requests never stop across
// the two MFMA phases between the tiles.
// TH_ST_NT = false: theta_new is stored with the default policy instead of non-temporally -- the NEXT step reads it twice (forward,
// update); while the online nets' Dense_0 kernels fit the memory-side cache beside the step's other traffic (K = 5: 79 MB of 256)
// they are found there (forward -4 us, update -2 ... -9 us per step; more heads: the dirty lines only get in the way, K = 8 +13 us
// -- the host chooses, qnet.hip d0_keep_heads).  m / v stay non-temporal unless ALL_DEFAULT.
template <bool ROWPAIR, int DEPTH = 4, bool TH_ST_NT = (D0_WG_NT & 2) != 0, bool ALL_DEFAULT = false>  // ALL_DEFAULT: every stream default-policy (K <= 2: theta, m, v and the target nets fit the memory-side cache together)
__device__ __forceinline__ void dense0_pair_body(const DenseWgradArgs& a, int item, float* gs /* 32 * 256 + 4096 + 1024 floats */, const int t) {
    constexpr int JT = 256, RPI = 4, NIT = 8;
    constexpr bool XT = DEPTH == NIT;  // cross-tile refills
    const int lane = t & 63, wave = t >> 6, bl = lane & 31, h = lane >> 5;
    int jt_fixed = 0;
    if (ROWPAIR) { jt_fixed = item & 1; item >>= 1; }
    const int nfp = ROWPAIR ? a.n_ft / 2 : a.n_ft;
    const int ft = (item % nfp) * (ROWPAIR ? 2 : 1), k = item / nfp, f0 = ft * 32, jw = wave * 64;
    const int prow = t >> 6, pcol = (t & 63) * 4;
    auto rot = [&](int row, int col) { return row * JT + ((col + 4 * row) & (JT - 1)); };
    const float bc1 = a.bcinv[2 * k], bc2 = a.bcinv[2 * k + 1];
    const float* const A3 = a.a3 + k * a.a3_head;  // (uniform bases + 32-bit byte offsets, as for theta / m / v below)
    const float* const Dh = a.dh + k * a.dh_head;
    auto tile_f0 = [&](int tt) { return ROWPAIR ? f0 + 32 * tt : f0; };
    auto tile_jt = [&](int tt) { return ROWPAIR ? jt_fixed : tt; };
    float4 x[4], y[2][4];
    float4 th[DEPTH], mm[DEPTH], vv[DEPTH];
    float dv[4][8];
    float4 mask4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_ops = [&](int tt) {
        const unsigned a_off = (unsigned)((tile_f0(tt) + bl) * 32 + 16 * h) * 4;
        const unsigned d_off = (unsigned)((tile_jt(tt) * JT + jw + bl) * 32 + 16 * h) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A3) + (a_off + 16 * u));
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                y[q][u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(Dh) + (d_off + (unsigned)(q * 32 * 32 * 4 + 16 * u)));
    };
    // theta / m / v: workgroup-uniform bases (scalar registers) + 32-bit element offsets, so that the sixteen row-group
    // addresses of the two tiles are adds on ONE register each, not 64-bit pointers held for the whole kernel
    const float* const Th = a.theta + (long)k * a.P + a.w_off;
    const float* const Mu = a.mu + (long)k * a.P + a.w_off;
    const float* const Nu = a.nu + (long)k * a.P + a.w_off;
    const unsigned rowJ = (unsigned)(RPI * a.J) * 4;  // BYTE offsets: a zero-extended 32-bit byte offset is what the scalar-base form takes
    auto at = [](const float* b, unsigned boff) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(b) + boff); };
    auto atw = [](const float* b, unsigned boff) { return reinterpret_cast<float*>(const_cast<char*>(reinterpret_cast<const char*>(b)) + boff); };
    auto prefetch = [&](unsigned o0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned on = o0 + (unsigned)d * rowJ;
            th[d] = ld4<!ALL_DEFAULT && (D0_WG_NT & 1) != 0>(at(Th, on));
            mm[d] = ld4<!ALL_DEFAULT && (D0_WG_NT & 1) != 0>(at(Mu, on));
            vv[d] = ld4<!ALL_DEFAULT && (D0_WG_NT & 1) != 0>(at(Nu, on));
        }
    };
    auto tile_base = [&](int tt) { return (unsigned)((tile_f0(tt) + prow) * a.J + tile_jt(tt) * JT + pcol) * 4; };
    load_ops(0);
    __builtin_amdgcn_sched_barrier(0);
    prefetch(tile_base(0));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int jt = tile_jt(tt);
        const unsigned o0 = tile_base(tt);
        // (the 2 x 32 LDS addresses of the park and of phase 3 are the same in both tiles: kept alive across the whole kernel
        // they cost 64 registers -- an opaque copy of the lane indices per tile makes hipcc recompute them)
        int blx = bl, hx = h;
        asm volatile("" : "+v"(blx), "+v"(hx));
        // phase 3, A operand: lane (b = bl, k = h) reads dh[j0 + jw + 2 u + h][b]
        const float* D3 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Dh) + (unsigned)((jt * JT + jw + h) * 32 + bl) * 4);
        // ---- phase 1: G = a3^T dh (f32 MFMA, k = the 32 samples in order) -> LDS, columns rotated by 4 * row
        {
            f32x16 acc[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
            const float av[16] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w,
                                  x[2].x, x[2].y, x[2].z, x[2].w, x[3].x, x[3].y, x[3].z, x[3].w};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float bv[16] = {y[q][0].x, y[q][0].y, y[q][0].z, y[q][0].w, y[q][1].x, y[q][1].y, y[q][1].z, y[q][1].w,
                                      y[q][2].x, y[q][2].y, y[q][2].z, y[q][2].w, y[q][3].x, y[q][3].y, y[q][3].z, y[q][3].w};
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[q] = mfma32(av[u], bv[u], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) gs[rot(mfma_row(r, hx), jw + 32 * q + blx)] = acc[q][r];
        }
        __syncthreads();
        // phase 3's first operand chunk goes out in front of the stream, the other three in its last iterations (into the
        // registers of ring slots that are not re-filled any more)
#pragma unroll
        for (int u = 0; u < 8; ++u) dv[0][u] = D3[(long)(2 * u) * 32];
        if (!ROWPAIR && tt == 1) {  // ... and so does the ReLU mask the finished rows need at the very end (else a round trip on its own)
            mask4 = *reinterpret_cast<const float4*>(A3 + (long)(f0 + (t >> 3)) * 32 + (t & 7) * 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- phase 2: streaming Adam, theta_old takes the consumed gradient's place in LDS
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int s = i % DEPTH;
            float* gp = &gs[rot(RPI * i + prow, pcol)];
            const float4 g = *reinterpret_cast<const float4*>(gp);
            float4 t4 = th[s], m4 = mm[s], v4 = vv[s];
            if (i >= NIT - 3) {
                const int c3 = 1 + i - (NIT - 3);
#pragma unroll
                for (int u = 0; u < 8; ++u) dv[c3][u] = D3[(long)(16 * c3 + 2 * u) * 32];
            }
            *reinterpret_cast<float4*>(gp) = t4;
            if (XT ? tt == 0 : i + DEPTH < NIT) {
                const unsigned on = XT ? tile_base(1) + (unsigned)i * rowJ : o0 + (unsigned)(i + DEPTH) * rowJ;
                th[s] = ld4<!ALL_DEFAULT && (D0_WG_NT & 1) != 0>(at(Th, on));
                mm[s] = ld4<!ALL_DEFAULT && (D0_WG_NT & 1) != 0>(at(Mu, on));
                vv[s] = ld4<!ALL_DEFAULT && (D0_WG_NT & 1) != 0>(at(Nu, on));
            }
            adam_elem(a.ad, bc1, bc2, g.x, t4.x, m4.x, v4.x);
            adam_elem(a.ad, bc1, bc2, g.y, t4.y, m4.y, v4.y);
            adam_elem(a.ad, bc1, bc2, g.z, t4.z, m4.z, v4.z);
            adam_elem(a.ad, bc1, bc2, g.w, t4.w, m4.w, v4.w);
            const unsigned o = o0 + (unsigned)i * rowJ;
            st4<TH_ST_NT>(atw(Th, o), t4);
            st4<!ALL_DEFAULT && (D0_WG_NT & 2) != 0>(atw(Mu, o), m4);
            st4<!ALL_DEFAULT && (D0_WG_NT & 2) != 0>(atw(Nu, o), v4);
        }
        __syncthreads();  // the LDS tile now holds theta_old[32][256] (rotated)
        if (tt == 0) {    // the second tile's contraction operands, in front of everything else it will request
            __builtin_amdgcn_sched_barrier(0);
            load_ops(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- phase 3: this wave's 64 columns of dL/da3[f][b] = sum_j theta_old[f][j] dh[j][b]
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int col = jw + 16 * c + 2 * u + hx;
                d = mfma32(dv[c][u], gs[blx * JT + ((col + 4 * blx) & (JT - 1))], d);
            }
        if (tt == 0 && !XT) {  // ... and its first four row groups, into the registers the phase-3 operands have left
            __builtin_amdgcn_sched_barrier(0);
            prefetch(tile_base(1));
            __builtin_amdgcn_sched_barrier(0);
        }
        float* red = gs + 32 * JT;  // [4 waves][32 f][32 b], 16-byte slots XOR-swizzled by (f & 7)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(&red[wave * 1024 + bl * 32 + (((2 * g + h) ^ (bl & 7)) * 4)]) =
                make_float4(d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]);
        __syncthreads();
        const int row = t >> 3, slot = ((t & 7) ^ (row & 7)) * 4;
        float4 s4 = *reinterpret_cast<const float4*>(&red[row * 32 + slot]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 yv = *reinterpret_cast<const float4*>(&red[w * 1024 + row * 32 + slot]);
            s4.x += yv.x; s4.y += yv.y; s4.z += yv.z; s4.w += yv.w;
        }
        if (ROWPAIR) {
            *reinterpret_cast<float4*>(a.dpart + ((long)jt * a.K + k) * a.F * 32 + (long)tile_f0(tt) * 32 + t * 4) = s4;
            if (tt == 0) __builtin_amdgcn_sched_barrier(0);
        } else if (tt == 0) {
            *reinterpret_cast<float4*>(&gs[32 * JT + 4096 + 4 * t]) = s4;  // kept in LDS (4 KB behind red) until the second tile's sum
            __builtin_amdgcn_sched_barrier(0);
        } else {
            const float4 part0 = *reinterpret_cast<const float4*>(&gs[32 * JT + 4096 + 4 * t]);
            // complete rows: tile 0 + tile 1, ReLU mask of a3, the three output forms of dL/da3 (as k_da3_finalize: one thread =
            // 4 samples of one row f, 8 threads a row)
            s4.x = part0.x + s4.x; s4.y = part0.y + s4.y; s4.z = part0.z + s4.z; s4.w = part0.w + s4.w;
            const int f = f0 + row, sl4 = (t & 7) * 4;
            const float4 m = mask4;
            s4.x = m.x > 0.f ? s4.x : 0.f; s4.y = m.y > 0.f ? s4.y : 0.f; s4.z = m.z > 0.f ? s4.z : 0.f; s4.w = m.w > 0.f ? s4.w : 0.f;
            const long sl = (long)k;
            const int pos = f / a.C, c = f - pos * a.C;
            const int oh = pos / a.g.W, ow = pos - oh * a.g.W;
            const long pix = (long)(oh + a.g.lo_h) * a.g.Wp + (ow + a.g.lo_w);
            if (a.da3f) *reinterpret_cast<float4*>(a.da3f + sl * a.g.block + (pix * a.C + c) * 32 + sl4) = s4;
            if (a.da3p) {
                unsigned short* O = a.da3p + sl * a.g.block * 3 + pix * (3L * a.C * 32) + (long)c * 32 + sl4;
                unsigned q0a, q1a, q2a, q0b, q1b, q2b;
                split3_pk(s4.x, s4.y, q0a, q1a, q2a);
                split3_pk(s4.z, s4.w, q0b, q1b, q2b);
                *reinterpret_cast<uint2*>(O) = make_uint2(q0a, q0b);
                *reinterpret_cast<uint2*>(O + (long)a.C * 32) = make_uint2(q1a, q1b);
                *reinterpret_cast<uint2*>(O + 2L * a.C * 32) = make_uint2(q2a, q2b);
            }
            if (a.pb) {
                float r = (s4.x + s4.y) + (s4.z + s4.w);
                r += __shfl_xor(r, 1);
                r += __shfl_xor(r, 2);
                r += __shfl_xor(r, 4);
                if ((t & 7) == 0) a.pb[(sl * (a.g.H * a.g.W) + pos) * a.C + c] = r;
            }
        }
    }
}


