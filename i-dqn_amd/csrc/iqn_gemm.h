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
//   forward        part[v][bb][s][j][b] = sum over the split's rows f of W[f][j] * x[v][bb][f][b]      (k_iqn_d0_fwd)
//   data gradient  dx[k][bb][f][b] = sum over j of W[f][j] * dh[k][bb][j][b]                             (k_iqn_d0_dgrad)
//   weight gradient g[ks][k][f][j] = sum over the blocks of split ks and b of x[k][bb][f][b] * dh[k][bb][j][b]  (k_iqn_d0_wgrad)
//                  followed by k_iqn_d0_adam: g = g[0] + g[1] (fixed order), Adam on Dense_0/kernel in one streaming pass
//   k_iqn_d0_bwd   both gradients in one launch (the default: 2 x 620 workgroups = 4.8 rounds of the chip instead of 3 + 3)
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
    long long* clk;  // diagnostic (IDQN_IQN_CLOCK=1) or nullptr: per workgroup {s_memtime, s_memrealtime} before and after the
                     // k loop -- the in-kernel clock is their ratio x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
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
    const long long c_entry = a.clk ? __builtin_amdgcn_s_memrealtime() : 0;
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
    long long c_t0 = 0, c_r0 = 0, d_lgkm = 0, d_bar = 0;
    if (a.clk) { c_t0 = __builtin_amdgcn_s_memtime(); c_r0 = __builtin_amdgcn_s_memrealtime(); }
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
                if (a.clk) {
                    const long long b0 = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const long long b1 = __builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_s_barrier();
                    const long long b2 = __builtin_amdgcn_s_memtime();
                    d_lgkm += b1 - b0; d_bar += b2 - b1;
                } else
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
    if (a.clk && threadIdx.x == 0) {
        long long* c = a.clk + (long)blockIdx.x * 4;
        c[0] = c_t0; c[1] = c_r0; c[2] = __builtin_amdgcn_s_memtime(); c[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (a.clk && lane == 0) {  // per wave: cycles waiting for its own LDS writes / in the barrier
        long long* c = a.clk + 1024 + ((long)blockIdx.x * 8 + wave) * 2;
        c[0] = d_lgkm; c[1] = d_bar;
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
    if (a.clk) {  // workgroup entry and the moment this wave's partial stores have left (100 MHz stamps)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            long long* c = a.clk + 1024 + 256 * 16 + (long)blockIdx.x * 2;
            c[0] = c_entry; c[1] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// ---- data gradient: dx = W0 . dh for every fraction block of the K online nets ------------------------------------------
// The per-block kernel (k_dense0_dgrad) runs on the f32 MFMA (1/16 of the bf16 rate; fine for the plain step, where it is
// bound by streaming W once): 0.48 ms for 32 blocks per net.  Same workgroup shape as the forward: a 256 (f rows) x 256
// (8 blocks) tile, k = the J hidden units in steps of 16; operand tiles: dh of one block (rows j, 8 dwords per lane, as the
// forward's activation tile) and 32 rows of W (16 consecutive floats of a lane's own row per k-step = 2 float4 per
// half-wave lane).  MFMA orientation as in k_dense0_dgrad (A = dh: rows = samples, B = W: columns = f): a lane ends up
// with ONE row f and 4 x 4 consecutive samples -> float4 stores.
struct IqnD0DgradArgs {
    const float* dh;            // [K][nb][J][32]
    const float* const* wbase;  // [K] online nets
    float* dx;                  // [K][nb][F][32]
    long w_off;
    int K, nb, F, J;
};

// Hand-off between the two gradients of one group = (head, 256 rows of Dense_0/kernel) when the Adam update rides in the weight
// gradient's epilogue (k_iqn_d0_bwd_adam): the data-gradient items of the group READ the rows the epilogue overwrites.  Each of
// them adds one arrival when its last kernel fragment has been consumed; a weight-gradient item waits for all of them before its
// first parameter store (bounded: it gives up after ~0.5 s, raises `err` -- the next steps' losses come out NaN -- and goes on),
// and the last of the group's weight-gradient items to pass re-arms both counters for the next step.
struct IqnD0Gate {
    unsigned* arrived;  // [groups]
    unsigned* passed;   // [groups]
    unsigned* err;
    long long* prof;    // debug build (IDQN_CONV_PROF=11): per workgroup {start, products done, wait done, end, item, -, -, -} (100 MHz clock)
};

// RT = row tiles per wave: a workgroup takes 64 RT rows of W (4: 256 rows, the GEMM tile of the other kernels; 3 / 2: more, shorter
// items -- the host picks what fills the chip, e.g. 205 items instead of 155 on 256 CUs for K = 5 heads x 8 blocks: qnet.hip)
template <int D, int RT = 4>
__device__ __forceinline__ void iqn_d0_dgrad_body(const IqnD0DgradArgs& a, int item, unsigned char* ig_lds, unsigned* arrive = nullptr) {
    constexpr int ROWS = 64 * RT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), bl = lane & 31, h = lane >> 5;
    // item = (net, ROWS-row f group, group of 8 blocks), blocks fastest: the workgroups that share a W row group are neighbours
    const int nbg = a.nb / 8, nfg = (a.F + ROWS - 1) / ROWS;
    const int bg = item % nbg;
    item /= nbg;
    const int fg = item % nfg;
    const int k = item / nfg;
    const int NC = a.J / 16;
    // producer role: W tile `wave` = rows f = 256 fg + 32 wave + bl (clamped past F: computed, not stored), lane (bl, h)
    // holds columns 16 c + 8 h .. + 7; dh tile `wave` = block 8 bg + wave, rows j = 16 c + 8 h + jj, sample bl
    const int frow = min(fg * ROWS + min(wave, 2 * RT - 1) * 32 + bl, a.F - 1);  // (RT < 4: waves 2 RT .. 7 re-park the last tile, nobody reads theirs)
    const float* Wp = a.wbase[k] + a.w_off + (long)frow * a.J + 8 * h;
    const float* Dp = a.dh + ((long)k * a.nb + bg * 8 + wave) * a.J * 32 + (long)(8 * h) * 32 + bl;
    // consumer role: W tiles RT wn .. RT wn + RT - 1 against blocks 2 wm, 2 wm + 1
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float wr[D][8], xr[D][8];
    auto fetch = [&](int slot, int c) {
        const float4 w0 = *reinterpret_cast<const float4*>(Wp + 16 * c), w1 = *reinterpret_cast<const float4*>(Wp + 16 * c + 4);
        wr[slot][0] = w0.x; wr[slot][1] = w0.y; wr[slot][2] = w0.z; wr[slot][3] = w0.w;
        wr[slot][4] = w1.x; wr[slot][5] = w1.y; wr[slot][6] = w1.z; wr[slot][7] = w1.w;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) xr[slot][jj] = Dp[(long)(16 * c + jj) * 32];
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, min(d, NC - 1));
    constexpr int U = (D % 2 == 0) ? D : 2 * D;
    unsigned char* const my_w = ig_lds + wave * IG_TILE + lane * 16;
    unsigned char* const my_x = ig_lds + (8 + wave) * IG_TILE + lane * 16;
    const unsigned char* const rd_w = ig_lds + (RT * wn) * IG_TILE + lane * 16;
    const unsigned char* const rd_x = ig_lds + (8 + 2 * wm) * IG_TILE + lane * 16;
    ig_park(wr[0], my_w);
    ig_park(xr[0], my_x);
    fetch(0, min(D, NC - 1));
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
                bf16x8 wf[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[0][p] = *LDS_PTR(const bf16x8, rd_w + stg + p * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // (four rounds of operand staging for the next k-step; products in the first RT of them)
                    if (i + 1 < RT) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) wf[(i + 1) & 1][p] = *LDS_PTR(const bf16x8, rd_w + stg + (i + 1) * IG_TILE + p * 1024);
                    }
                    const bf16x8 w0 = wf[i & 1][0], w1 = wf[i & 1][1], w2 = wf[i & 1][2];
                    __builtin_amdgcn_sched_barrier(0);
                    if (i < RT) {
                    acc[i][0] = mfma_bf16(xf[0][2], w0, acc[i][0]);
                    acc[i][1] = mfma_bf16(xf[1][2], w0, acc[i][1]);
                    acc[i][0] = mfma_bf16(xf[0][0], w2, acc[i][0]);
                    acc[i][1] = mfma_bf16(xf[1][0], w2, acc[i][1]);
                    acc[i][0] = mfma_bf16(xf[0][1], w1, acc[i][0]);
                    acc[i][1] = mfma_bf16(xf[1][1], w1, acc[i][1]);
                    }
                    {
                        const int m = (2 * i) & 3;
                        if (i < 2) split3_pk(wr[nslot][2 * m], wr[nslot][2 * m + 1], p0[m], p1[m], p2[m]);
                        else split3_pk(xr[nslot][2 * m], xr[nslot][2 * m + 1], p0[m], p1[m], p2[m]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (i < RT) {
                    acc[i][0] = mfma_bf16(xf[0][1], w0, acc[i][0]);
                    acc[i][1] = mfma_bf16(xf[1][1], w0, acc[i][1]);
                    acc[i][0] = mfma_bf16(xf[0][0], w1, acc[i][0]);
                    acc[i][1] = mfma_bf16(xf[1][0], w1, acc[i][1]);
                    acc[i][0] = mfma_bf16(xf[0][0], w0, acc[i][0]);
                    acc[i][1] = mfma_bf16(xf[1][0], w0, acc[i][1]);
                    }
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
                fetch(nslot, min(c + 1 + D, NC - 1));
            }
        }
    }
    // this lane of tile (i, j): row f = ROWS fg + 32 (RT wn + i) + bl, samples (r & 3) + 8 (r >> 2) + 4 h of block 8 bg + 2 wm + j
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int f = fg * ROWS + (RT * wn + i) * 32 + bl;
        if (f >= a.F) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* O = a.dx + (((long)k * a.nb + bg * 8 + 2 * wm + j) * a.F + f) * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(O + 8 * g) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
    }
    if (arrive) {  // (every wave is past its last product: the kernel fragments it asked for and used have arrived)
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------
// The plain step's fused kernel (dense0_update.h) contracts the sample blocks inside the workgroups that stream theta / m / v:
// one 32 x 256 tile each, every factor fragment fetched per tile -- right for 1..8 blocks (HBM-bound), 0.41 + 0.07 ms at 32
// blocks (the fragment traffic of 980 x 32 tile-blocks).  Here: the same 256 x 256 GEMM tile (f rows x j columns), k = the
// samples of the blocks, 16 per k-step (two k-steps per block); both operands are k-contiguous in memory (a lane's 8 samples
// of its own row = 2 float4).  The block range is cut in `KS` splits so that K x 31 x 2 x KS workgroups fill the chip
// more evenly; each split writes its own partial, k_iqn_d0_adam adds them in split order (reproducible) inside the Adam pass.
struct IqnD0WgradArgs {
    const float* x;    // [K][nb][F][32]  (the online virtual nets come first in xq)
    const float* dh;   // [K][nb][J][32]
    float* g[2];       // partial sums [K][F][J] of split 0 / 1
    int K, nb, F, J, KS;
    // ADAM (KS = 1: the tile's sum is complete): the epilogue updates Dense_0/kernel itself -- parameter arenas with head stride P,
    // the kernel at w_off; no gradient leaves the registers
    float *theta, *mu, *nu;
    const float* bcinv;
    AdamConsts ad;
    long P, w_off;
    const char* dump;  // >= 2 KB + 12 rows of J floats: what the epilogue of a tile past row F reads and writes
};

template <int D, int RT = 4>
__global__ __launch_bounds__(512) void k_iqn_d0_dgrad(IqnD0DgradArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ig_lds[];
    iqn_d0_dgrad_body<D, RT>(a, xcd_contiguous_id(), ig_lds);
}


template <int D, bool ADAM = false>
__device__ __forceinline__ void iqn_d0_wgrad_body(const IqnD0WgradArgs& a, int item, unsigned char* ig_lds, const IqnD0Gate* gate = nullptr, int group = 0, int n_wait = 0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), bl = lane & 31, h = lane >> 5;
    const int n_jh = a.J / 256, nfg = (a.F + 255) / 256;
    const int jh = item % n_jh;
    item /= n_jh;
    const int ks = item % a.KS;
    item /= a.KS;
    const int fg = item % nfg;
    const int k = item / nfg;
    const int nbs = a.nb / a.KS, b0 = ks * nbs, NC = 2 * nbs;  // k-step c: block b0 + c / 2, samples 16 (c & 1) + 8 h .. + 7
    // producer role: x tile `wave` = rows f = 256 fg + 32 wave + bl (clamped past F), dh tile `wave` = columns j = 256 jh + 32 wave + bl
    const int frow = min(fg * 256 + wave * 32 + bl, a.F - 1);
    const float* Xp = a.x + ((long)k * a.nb + b0) * a.F * 32 + (long)frow * 32 + 8 * h;
    const float* Dp = a.dh + ((long)k * a.nb + b0) * a.J * 32 + (long)(jh * 256 + wave * 32 + bl) * 32 + 8 * h;
    const long xblk = (long)a.F * 32, dblk = (long)a.J * 32;
    // consumer role: x tiles 4 wn .. 4 wn + 3 (rows) against dh tiles 2 wm, 2 wm + 1 (columns)
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float wr[D][8], xr[D][8];
    auto fetch = [&](int slot, int c) {
        const long ob = (long)(c >> 1), oh = 16 * (c & 1);
        const float4 x0 = *reinterpret_cast<const float4*>(Xp + ob * xblk + oh), x1 = *reinterpret_cast<const float4*>(Xp + ob * xblk + oh + 4);
        const float4 d0 = *reinterpret_cast<const float4*>(Dp + ob * dblk + oh), d1 = *reinterpret_cast<const float4*>(Dp + ob * dblk + oh + 4);
        wr[slot][0] = x0.x; wr[slot][1] = x0.y; wr[slot][2] = x0.z; wr[slot][3] = x0.w;
        wr[slot][4] = x1.x; wr[slot][5] = x1.y; wr[slot][6] = x1.z; wr[slot][7] = x1.w;
        xr[slot][0] = d0.x; xr[slot][1] = d0.y; xr[slot][2] = d0.z; xr[slot][3] = d0.w;
        xr[slot][4] = d1.x; xr[slot][5] = d1.y; xr[slot][6] = d1.z; xr[slot][7] = d1.w;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, min(d, NC - 1));
    constexpr int U = (D % 2 == 0) ? D : 2 * D;
    unsigned char* const my_w = ig_lds + wave * IG_TILE + lane * 16;
    unsigned char* const my_x = ig_lds + (8 + wave) * IG_TILE + lane * 16;
    const unsigned char* const rd_w = ig_lds + (4 * wn) * IG_TILE + lane * 16;
    const unsigned char* const rd_x = ig_lds + (8 + 2 * wm) * IG_TILE + lane * 16;
    ig_park(wr[0], my_w);
    ig_park(xr[0], my_x);
    fetch(0, min(D, NC - 1));
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
                bf16x8 wf[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[0][p] = *LDS_PTR(const bf16x8, rd_w + stg + p * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < 3) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) wf[(i + 1) & 1][p] = *LDS_PTR(const bf16x8, rd_w + stg + (i + 1) * IG_TILE + p * 1024);
                    }
                    const bf16x8 w0 = wf[i & 1][0], w1 = wf[i & 1][1], w2 = wf[i & 1][2];
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][0] = mfma_bf16(w2, xf[0][0], acc[i][0]);
                    acc[i][1] = mfma_bf16(w2, xf[1][0], acc[i][1]);
                    acc[i][0] = mfma_bf16(w0, xf[0][2], acc[i][0]);
                    acc[i][1] = mfma_bf16(w0, xf[1][2], acc[i][1]);
                    acc[i][0] = mfma_bf16(w1, xf[0][1], acc[i][0]);
                    acc[i][1] = mfma_bf16(w1, xf[1][1], acc[i][1]);
                    {
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
                fetch(nslot, min(c + 1 + D, NC - 1));
            }
        }
    }
    // tile (i, j): rows f = 256 fg + 32 (4 wn + i) + mfma_row(r, h), column j = 256 jh + 32 (2 wm + j) + bl: a store
    // instruction writes two 128-byte row pieces
    if constexpr (ADAM) {
        if (gate) {
            if (threadIdx.x == 0) {
                if (gate->prof) gate->prof[8L * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
                int polls = 0;
                while (__hip_atomic_load(gate->arrived + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n_wait) {
                    if (++polls > (1 << 18)) { __hip_atomic_store(gate->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(64);
                }
                const unsigned old = __hip_atomic_fetch_add(gate->passed + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)n_jh - 1u) {  // both have read `arrived`: re-armed for the next step
                    __hip_atomic_store(gate->arrived + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(gate->passed + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (gate->prof) gate->prof[8L * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
            }
            __syncthreads();
        }
        // Every stream non-temporal, uniform bases + 32-bit byte offsets (the kernel of one head is < 4 GB).  A wave's 8 tiles are
        // walked as 16 chunks of 8 rows (half a 32 x 32 tile: 3 x 8 loads per lane), EPI_NB - 1 chunks requested ahead of the one
        // being updated: tile by tile -- 48 loads, wait, 48 stores, and the next loads queued behind those stores -- an item spent
        // 54 us here, one full round trip per tile.
        if (!a.theta) return;  // (timing probe of the debug build: the schedule without the epilogue)
        const float bc1 = a.bcinv[2 * k], bc2 = a.bcinv[2 * k + 1];
        const long hb = (long)k * a.P + a.w_off;
        const char* const T = reinterpret_cast<const char*>(a.theta + hb);
        const char* const M = reinterpret_cast<const char*>(a.mu + hb);
        const char* const V = reinterpret_cast<const char*>(a.nu + hb);
        const unsigned rowb = (unsigned)a.J * 4u;
        constexpr int EPI_NB = 4;
        float th[EPI_NB][8], mm[EPI_NB][8], vv[EPI_NB][8];
        // chunk c = (tile row i = c >> 2, tile column j = (c >> 1) & 1, row half c & 1): accumulator registers 8 (c & 1) + rr, i.e.
        // rows mfma_row(8 (c & 1) + rr, h) = 16 (c & 1) + 8 (rr >> 2) + (rr & 3) + 4 h
        auto live = [&](int c) { return fg * 256 + (4 * wn + (c >> 2)) * 32 < a.F; };  // (F is a multiple of 32: whole tiles are in or out)
        auto cofs = [&](int c) {
            const int f0 = fg * 256 + (4 * wn + (c >> 2)) * 32 + 16 * (c & 1) + 4 * h;
            return (unsigned)f0 * rowb + (unsigned)(jh * 256 + (2 * wm + ((c >> 1) & 1)) * 32 + bl) * 4u;
        };
        // (no branches in here: across basic blocks hipcc drains vmcnt to 0 in front of every store.  The tiles past row F of the
        // last row group read and write a 32 KB dump instead -- a.dump, never read by anyone)
        auto request = [&](int c) {
            const bool lv = live(c);
            const unsigned o0 = lv ? cofs(c) : threadIdx.x * 4u;
            const char *Tc = lv ? T : a.dump, *Mc = lv ? M : a.dump, *Vc = lv ? V : a.dump;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const unsigned o = o0 + (unsigned)(8 * (rr >> 2) + (rr & 3)) * rowb;
                th[c % EPI_NB][rr] = __builtin_nontemporal_load(reinterpret_cast<const float*>(Tc + o));
                mm[c % EPI_NB][rr] = __builtin_nontemporal_load(reinterpret_cast<const float*>(Mc + o));
                vv[c % EPI_NB][rr] = __builtin_nontemporal_load(reinterpret_cast<const float*>(Vc + o));
            }
        };
#pragma unroll
        for (int c = 0; c < EPI_NB - 1; ++c) request(c);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c + EPI_NB - 1 < 16) request(c + EPI_NB - 1);
            const bool lv = live(c);
            const unsigned o0 = lv ? cofs(c) : threadIdx.x * 4u;
            char *Tc = const_cast<char*>(lv ? T : a.dump), *Mc = const_cast<char*>(lv ? M : a.dump), *Vc = const_cast<char*>(lv ? V : a.dump);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const unsigned o = o0 + (unsigned)(8 * (rr >> 2) + (rr & 3)) * rowb;
                float t1 = th[c % EPI_NB][rr], m1 = mm[c % EPI_NB][rr], v1 = vv[c % EPI_NB][rr];
                adam_elem(a.ad, bc1, bc2, acc[c >> 2][(c >> 1) & 1][8 * (c & 1) + rr], t1, m1, v1);
                __builtin_nontemporal_store(t1, reinterpret_cast<float*>(Tc + o));
                __builtin_nontemporal_store(m1, reinterpret_cast<float*>(Mc + o));
                __builtin_nontemporal_store(v1, reinterpret_cast<float*>(Vc + o));
            }
        }
        return;
    }
    float* G = a.g[ks] + (long)k * a.F * a.J + jh * 256 + bl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f0 = fg * 256 + (4 * wn + i) * 32;
        if (f0 >= a.F) continue;  // F is a multiple of 32: whole tiles are in or out
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) G[(long)(f0 + mfma_row(r, h)) * a.J + (2 * wm + j) * 32] = acc[i][j][r];
    }
}

template <int D>
__global__ __launch_bounds__(512) void k_iqn_d0_wgrad(IqnD0WgradArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ig_lds[];
    iqn_d0_wgrad_body<D>(a, xcd_contiguous_id(), ig_lds);
}

// Both gradients in ONE launch: they are independent (dx needs W and dh, g needs x and dh), each has 620 equal workgroups at
// K = 5 -- 2.4 rounds of the 256 CUs, i.e. three rounds with the last one 42 % full; together 4.8 rounds, five.  The first
// n_dgrad workgroups take data-gradient items, the rest weight-gradient items.
template <int D>
__global__ __launch_bounds__(512) void k_iqn_d0_bwd(IqnD0DgradArgs d, int n_dgrad, IqnD0WgradArgs w) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ig_lds[];
    if ((int)blockIdx.x < n_dgrad) {
        iqn_d0_dgrad_body<D>(d, xcd_contiguous_id_n(n_dgrad), ig_lds);
    } else {
        const int n = (int)gridDim.x - n_dgrad, b = (int)blockIdx.x - n_dgrad;
        const int q = n >> 3, r = n & 7, x = b & 7;
        iqn_d0_wgrad_body<D>(w, (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3), ig_lds);
    }
}

// The same launch with the Adam update of Dense_0/kernel in the weight gradient's epilogue (one split: the tile's sum is complete in
// registers): no gradient written and read back (2 x 79 MB at K = 5 with one split), no Adam launch, and the parameter streams of a
// finishing item run under the other CUs' matrix work.  A group = (head, 256 kernel rows): nb / 8 data-gradient items, which read
// those rows, and J / 256 weight-gradient items, which overwrite them (IqnD0Gate).  Group g lives on XCD g % 8 (workgroup b runs on
// XCD b % 8), its data-gradient items at earlier slots there than its weight-gradient items: they are dispatched before the items
// that wait for them and wait for nothing themselves, so the wait cannot block progress, and they share the group's kernel rows /
// x rows through that XCD's L2.  The order itself is the host's (qnet.hip, plan_iqn_bwd_order: it decides where the long items
// -- two k-halves + the epilogue -- start, i.e. the tail of the launch).
template <int D>
__global__ __launch_bounds__(512) void k_iqn_d0_bwd_adam(IqnD0DgradArgs d, IqnD0WgradArgs w, IqnD0Gate gate, const int32_t* __restrict__ items) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ig_lds[];
    const int nbg = d.nb / 8, n_jh = w.J / 256;
    const int e = __builtin_amdgcn_readfirstlane(items[blockIdx.x]);  // (group << 8) | item of the group, or -1 (qnet.hip, plan_iqn_bwd_order)
    if (e < 0) return;
    const int group = e >> 8, within = e & 255;
    if (gate.prof && threadIdx.x == 0) { gate.prof[8L * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); gate.prof[8L * blockIdx.x + 4] = e; }
    if (within < nbg) iqn_d0_dgrad_body<D>(d, group * nbg + within, ig_lds, gate.arrived + group);
    else iqn_d0_wgrad_body<D, true>(w, group * n_jh + (within - nbg), ig_lds, &gate, group, nbg);
    if (gate.prof) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) gate.prof[8L * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// Adam on Dense_0/kernel from the partial gradient sums: one float4 per thread, every stream non-temporal (dense0_update.h)
struct IqnD0AdamArgs {
    const float* g[2];
    float *theta, *mu, *nu;  // parameter arenas, head stride P, Dense_0/kernel at w_off
    const float* bcinv;
    AdamConsts ad;
    long P, w_off, n;        // n = F * J
    int KS;
};
__global__ __launch_bounds__(256) void k_iqn_d0_adam(IqnD0AdamArgs a) {
    const int k = blockIdx.y;
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= a.n) return;
    const long o = (long)k * a.P + a.w_off + e;
    float4 th = ld4<true>(a.theta + o), m = ld4<true>(a.mu + o), v = ld4<true>(a.nu + o);
    float4 g = ld4<true>(a.g[0] + (long)k * a.n + e);
    if (a.KS > 1) {
        const float4 g1 = ld4<true>(a.g[1] + (long)k * a.n + e);
        g.x += g1.x; g.y += g1.y; g.z += g1.z; g.w += g1.w;
    }
    const float bc1 = a.bcinv[2 * k], bc2 = a.bcinv[2 * k + 1];
    adam_elem(a.ad, bc1, bc2, g.x, th.x, m.x, v.x);
    adam_elem(a.ad, bc1, bc2, g.y, th.y, m.y, v.y);
    adam_elem(a.ad, bc1, bc2, g.z, th.z, m.z, v.z);
    adam_elem(a.ad, bc1, bc2, g.w, th.w, m.w, v.w);
    st4<true>(a.theta + o, th);
    st4<true>(a.mu + o, m);
    st4<true>(a.nu + o, v);
}
