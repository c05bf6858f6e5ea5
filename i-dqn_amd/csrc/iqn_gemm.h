// Dense_0 of the i-IQN heads as tiled GEMMs on the bf16 matrix cores (f32 accuracy: convp.h's three-plane split, six
// products).  Included by iqn_kernels.h.
//
// Why.  The plain step gives Dense_0 one wave per (net, 32-sample block): W is streamed once per net and step and the
// kernels are HBM-bound (k_dense0_fwd3, k_dense0_dgrad).  With the quantile heads a virtual net sees N = 32 blocks per
// step (1024 rows), the same kernels fetch and split every element of W 32 times and every activation 4 times
// (0.94 ms forward at K = 5: bound by the bytes the CUs keep in flight, the MFMAs idle 2/3 of the time).  Here the
// contraction is a real GEMM: a workgroup (8 waves) owns a 256 x 256 output tile -- 256 columns of W x 8 blocks -- and per
// 16-row k-step every wave fetches ONE 32-wide operand tile (a wave-row of W or of one block's activations), splits it
// into the three bf16 planes once, and parks the fragments -- exactly the 16-byte vectors the MFMA takes -- in LDS; after
// one barrier each wave runs its 4 x 2 tiles (48 MFMAs) against fragments read back from LDS.  Per k-step and workgroup:
// 32 KB fetched and 128 element-pairs split per lane-pair for 384 MFMAs (the per-block kernel: 80 KB and 4x the split work).
//   forward   part[v][bb][s][j][b] = sum over the split's rows f of W[f][j] * x[v][bb][f][b]      (k_iqn_d0_fwd)
// The k-steps of a split are taken in the same order and the six products in the same order as in k_dense0_fwd3: the
// partials are bit-identical to that kernel's.
#pragma once
#include "cnn_kernels.h"

struct IqnD0FwdArgs {
    const float* x;             // [V][nb][F][32]
    const float* const* wbase;  // [V]
    float* part;                // [V][nb][NS][J][32]
    long w_off;
    int V, nb, NS, F, J;
};

constexpr int IG_TILE = 3 * 1024;        // one operand tile of a k-step: 3 planes x 64 lanes x 16 B
constexpr int IG_STAGE = 16 * IG_TILE;   // 8 W tiles + 8 activation tiles

__device__ __forceinline__ void ig_park(const float (&v)[8], unsigned char* dst) {
    unsigned p0[4], p1[4], p2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) split3_pk(v[2 * m], v[2 * m + 1], p0[m], p1[m], p2[m]);
    *LDS_PTR(u32x4, dst) = (u32x4){p0[0], p0[1], p0[2], p0[3]};
    *LDS_PTR(u32x4, dst + 1024) = (u32x4){p1[0], p1[1], p1[2], p1[3]};
    *LDS_PTR(u32x4, dst + 2048) = (u32x4){p2[0], p2[1], p2[2], p2[3]};
}

template <int D>  // k-steps of operand rows in flight per wave (registers)
__global__ __launch_bounds__(512) void k_iqn_d0_fwd(IqnD0FwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ig_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), bl = lane & 31, h = lane >> 5;
    // item = (net, group of 8 blocks, split, 256-column half), net slowest; an XCD walks consecutive items, i.e. the
    // workgroups that stream the same net's kernel share it through one L2
    int item = xcd_contiguous_id();
    const int n_jh = a.J / 256, nbg = a.nb / 8;
    const int jh = item % n_jh;
    item /= n_jh;
    const int s = item % a.NS;
    item /= a.NS;
    const int bg = item % nbg;
    const int n = item / nbg;
    const int NU = a.F / 16, NC = (NU - s + a.NS - 1) / a.NS;
    const long step_rows = 16L * a.NS;
    // producer role: W tile `wave` = columns jh * 256 + 32 wave + bl, activation tile `wave` = block 8 bg + wave; lane
    // (bl, h) holds rows 8 h .. 8 h + 7 of the k-step (the MFMA's k index) of its column
    const float* Wp = a.wbase[n] + a.w_off + (long)(16 * s + 8 * h) * a.J + jh * 256 + wave * 32 + bl;
    const float* Xp = a.x + ((long)n * a.nb + bg * 8 + wave) * a.F * 32 + (long)(16 * s + 8 * h) * 32 + bl;
    const long wstep = step_rows * a.J, xstep = step_rows * 32;
    // consumer role: W tiles 4 wn .. 4 wn + 3 against blocks 2 wm, 2 wm + 1
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // Software pipeline.  Iteration c: one barrier (stage c & 1, parked during iteration c - 1, is complete; nobody reads
    // the other stage any more), then the 48 MFMAs of k-step c with the split of k-step c + 1 threaded between them -- one
    // split3_pk (14 VALU instructions) per six MFMAs rides in the shadow of the matrix pipe -- and its twelve fragment
    // vectors written to the other stage.  Registers hold k-steps c + 1 .. c + D of this wave's two operand tiles.
    float wr[D][8], xr[D][8];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const long c = min(d, NC - 1);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            wr[d][jj] = Wp[c * wstep + (long)jj * a.J];
            xr[d][jj] = Xp[c * xstep + jj * 32];
        }
    }
    constexpr int U = (D % 2 == 0) ? D : 2 * D;
    unsigned char* const my_w = ig_lds + wave * IG_TILE + lane * 16;
    unsigned char* const my_x = ig_lds + (8 + wave) * IG_TILE + lane * 16;
    const unsigned char* const rd_w = ig_lds + (4 * wn) * IG_TILE + lane * 16;
    const unsigned char* const rd_x = ig_lds + (8 + 2 * wm) * IG_TILE + lane * 16;
    ig_park(wr[0], my_w);
    ig_park(xr[0], my_x);
    {   // slot 0 is free again: k-step D
        const long cn = min(D, NC - 1);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            wr[0][jj] = Wp[cn * wstep + (long)jj * a.J];
            xr[0][jj] = Xp[cn * xstep + jj * 32];
        }
    }
    for (int c0 = 0; c0 < NC; c0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u;
            if (c < NC) {  // (uniform over the workgroup)
                const int nslot = (u + 1) % D, stg = (u & 1) * IG_STAGE, nstg = ((u + 1) & 1) * IG_STAGE;
                lds_barrier();
                bf16x8 xf[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) xf[j][p] = *LDS_PTR(const bf16x8, rd_x + stg + j * IG_TILE + p * 1024);
                unsigned p0[4], p1[4], p2[4];
                bf16x8 wf[2][3];  // the W fragments of tile i + 1 are requested before the products of tile i are issued
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[0][p] = *LDS_PTR(const bf16x8, rd_w + stg + p * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < 3) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) wf[(i + 1) & 1][p] = *LDS_PTR(const bf16x8, rd_w + stg + (i + 1) * IG_TILE + p * 1024);
                    }
                    const bf16x8 w0 = wf[i & 1][0], w1 = wf[i & 1][1], w2 = wf[i & 1][2];
                    // the two blocks' accumulators take turns (each still sees its six products in the same order): a
                    // product never waits for the one issued just before it
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][0] = mfma_bf16(w2, xf[0][0], acc[i][0]);
                    acc[i][1] = mfma_bf16(w2, xf[1][0], acc[i][1]);
                    acc[i][0] = mfma_bf16(w0, xf[0][2], acc[i][0]);
                    acc[i][1] = mfma_bf16(w0, xf[1][2], acc[i][1]);
                    acc[i][0] = mfma_bf16(w1, xf[0][1], acc[i][0]);
                    acc[i][1] = mfma_bf16(w1, xf[1][1], acc[i][1]);
                    {   // pairs 2 i, 2 i + 1 (mod 4) of the next k-step's rows: i < 2 the W tile, else the activation tile
                        const int m = (2 * i) & 3;
                        if (i < 2) split3_pk(wr[nslot][2 * m], wr[nslot][2 * m + 1], p0[m], p1[m], p2[m]);
                        else split3_pk(xr[nslot][2 * m], xr[nslot][2 * m + 1], p0[m], p1[m], p2[m]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][0] = mfma_bf16(w1, xf[0][0], acc[i][0]);
                    acc[i][1] = mfma_bf16(w1, xf[1][0], acc[i][1]);
                    acc[i][0] = mfma_bf16(w0, xf[0][1], acc[i][0]);
                    acc[i][1] = mfma_bf16(w0, xf[1][1], acc[i][1]);
                    acc[i][0] = mfma_bf16(w0, xf[0][0], acc[i][0]);
                    acc[i][1] = mfma_bf16(w0, xf[1][0], acc[i][1]);
                    {
                        const int m = (2 * i + 1) & 3;
                        if (i < 2) split3_pk(wr[nslot][2 * m], wr[nslot][2 * m + 1], p0[m], p1[m], p2[m]);
                        else split3_pk(xr[nslot][2 * m], xr[nslot][2 * m + 1], p0[m], p1[m], p2[m]);
                        if (m == 3) {
                            unsigned char* dst = (i < 2 ? my_w : my_x) + nstg;
                            *LDS_PTR(u32x4, dst) = (u32x4){p0[0], p0[1], p0[2], p0[3]};
                            *LDS_PTR(u32x4, dst + 1024) = (u32x4){p1[0], p1[1], p1[2], p1[3]};
                            *LDS_PTR(u32x4, dst + 2048) = (u32x4){p2[0], p2[1], p2[2], p2[3]};
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // the slot just parked takes k-step c + 1 + D
                    const long cn = min(c + 1 + D, NC - 1);
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        wr[nslot][jj] = Wp[cn * wstep + (long)jj * a.J];
                        xr[nslot][jj] = Xp[cn * xstep + jj * 32];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int bb = bg * 8 + 2 * wm + j;
        float* P = a.part + (((long)n * a.nb + bb) * a.NS + s) * a.J * 32 + bl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col0 = jh * 256 + (4 * wn + i) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) P[(long)(col0 + mfma_row(r, h)) * 32] = acc[i][j][r];
        }
    }
}
