// gfx950 kernels of the i-IQN heads (BASELINE config 3) -- a LABELLED EXTENSION, parity unpinned: the reference snapshot
// has no quantile code (/root/reference/README.md:3,10 only names i-IQN); oracle/iqn_ref.py states the algorithm
// (implicit quantile network of Dabney et al. 2018 on the reference's conv trunk, the reference's chain of K heads) and its
// sources.  Included by qnet.hip.
//
// Mapping onto the existing step: a (sample, quantile fraction) pair is one more "sample"; quantile index q of all 32
// samples of the batch is one more 32-sample BLOCK, so the layouts of the plain step carry over to V = 3K "virtual nets"
// (online k | target k for the action choice | target k for the values) x N blocks each:
//   x[v][q][f][b] = psi[net(v)][f][b] * relu(sum_i cos(pi i tau[v][q][b]) * We[i][f] + be[f])   (k_iqn_cos, k_iqn_we_pack, k_iqn_embed3l)
//   Dense_0 over the (V, N) blocks: the tiled GEMM k_iqn_d0_fwd (iqn_gemm.h; fewer than 8 blocks: k_dense0_fwd3), k_hidden
//   k_iqn_z, k_iqn_loss   Z, a*, targets, the N' x N quantile Huber loss and dL/dZ                (one workgroup per head)
//   k_iqn_dh              dL/dh per block, Dense_1 / Dense_0-bias gradients (fraction groups, k_iqn_head_grad_sum)
//   k_iqn_d0_bwd_adam     W0 . dh (plain rows) and the Dense_0 weight gradient as GEMMs in one launch, Adam of Dense_0/kernel in the
//                         weight gradient's epilogue (few items: k_iqn_d0_bwd with two block splits + k_iqn_d0_adam)
//   k_iqn_embed_bwd3      dL/dpsi (summed over the N fractions) and the embedding's gradients (k_iqn_embed_grad_sum)
// then the plain step's conv backward and small-leaf Adam.  N not a multiple of 8 / 16: the plain step's per-block Dense_0 kernels.
#pragma once
#include "cnn_kernels.h"
#include "iqn_gemm.h"

constexpr int IQN_EMBED = 64;  // cos features per fraction (IQN paper, section 3; Dopamine quantile_embedding_dim)

// cos(pi * i * tau), i = 1..64, evaluated in fp64 and rounded once.  tau: [K][3][N][B] (floats in
// (0, 1)); slot = (type * K + k) * N + q.  Padded samples (b >= B) get tau = 0.5.
struct IqnCosArgs {
    const float* tau;
    unsigned short* cosp;  // the block as MFMA B fragments in three exact bf16 planes: cosp[slot][k-step t][plane][lane (b, h)][8]
                           // = cos feature i = 16 t + 8 h + jj of sample b (k_iqn_embed3), or nullptr
    unsigned short* cosa;  // ... and as A fragments of cos (rows = features, k = samples): cosa[slot][row tile rt][k-step t][plane]
                           // [lane (i, h)][8] = feature 32 rt + i of samples 16 t + 8 h + jj (k_iqn_embed_bwd3), or nullptr
    int K, N, B;
};
__global__ __launch_bounds__(256) void k_iqn_cos(IqnCosArgs a) {
    __shared__ float ct[IQN_EMBED][33];
    const int slot = blockIdx.x, q = slot % a.N, v = slot / a.N, type = v / a.K, k = v - type * a.K;
    const int b = threadIdx.x & 31;
    const double tau = b < a.B ? (double)a.tau[(((long)k * 3 + type) * a.N + q) * a.B + b] : 0.5;
    const int g = threadIdx.x >> 5;  // this thread: features i = 8 g .. 8 g + 7 = k-step g >> 1, half g & 1
    float c[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int i = 8 * g + jj;
        c[jj] = (float)cospi((double)(i + 1) * tau);
        ct[i][b] = c[jj];
    }
    if (a.cosp) {
        unsigned p0[4], p1[4], p2[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) split3_pk(c[2 * m], c[2 * m + 1], p0[m], p1[m], p2[m]);
        unsigned short* O = a.cosp + (((long)slot * 4 + (g >> 1)) * 3) * 512 + ((g & 1) * 32 + b) * 8;
        *reinterpret_cast<u32x4*>(O) = (u32x4){p0[0], p0[1], p0[2], p0[3]};
        *reinterpret_cast<u32x4*>(O + 512) = (u32x4){p1[0], p1[1], p1[2], p1[3]};
        *reinterpret_cast<u32x4*>(O + 1024) = (u32x4){p2[0], p2[1], p2[2], p2[3]};
    }
    if (a.cosa) {  // thread = (row tile rt, k-step t, lane (i, h)): 8 consecutive samples of one feature
        __syncthreads();
        const int lane = threadIdx.x & 63, il = lane & 31, hh = lane >> 5, rt = threadIdx.x >> 7, t = (threadIdx.x >> 6) & 1;
        float v8[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v8[jj] = ct[32 * rt + il][16 * t + 8 * hh + jj];
        unsigned p0[4], p1[4], p2[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) split3_pk(v8[2 * m], v8[2 * m + 1], p0[m], p1[m], p2[m]);
        unsigned short* O = a.cosa + (((long)slot * 4 + rt * 2 + t) * 3) * 512 + lane * 8;
        *reinterpret_cast<u32x4*>(O) = (u32x4){p0[0], p0[1], p0[2], p0[3]};
        *reinterpret_cast<u32x4*>(O + 512) = (u32x4){p1[0], p1[1], p1[2], p1[3]};
        *reinterpret_cast<u32x4*>(O + 1024) = (u32x4){p2[0], p2[1], p2[2], p2[3]};
    }
}

// ---- the embedding on the bf16 matrix cores (f32 accuracy: three exact planes, six products -- convp.h) -------------------
// x = psi * relu(We^T cos + be): one wave = one tile of 32 features x 32 samples of one (virtual net, fraction) block.  As 32 f32
// MFMAs (2048 cycles) per tile the K = 5 step spent 0.20 ms here, MFMA-bound.  Both operands are small and shared by many tiles, so they are split ONCE per step into fragment-ordered planes (k_iqn_cos writes the cos
// blocks, k_iqn_we_pack the embedding kernels) and the tile costs 24 bf16 MFMAs (768 cycles) and no split work.
struct IqnWePackArgs {
    const float* const* wbase;  // [n_nets]
    unsigned short* wep;        // [n_nets][F / 32][k-step t][plane][lane (f, h)][8] = We[16 t + 8 h + jj][32 ft + f]
    long we_off;
    int F;
};
__global__ __launch_bounds__(256) void k_iqn_we_pack(IqnWePackArgs a) {
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6, bl = lane & 31, h = lane >> 5;
    const int ft = blockIdx.x, net = blockIdx.y;
    const float* We = a.wbase[net] + a.we_off + (long)(16 * t + 8 * h) * a.F + ft * 32 + bl;
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) v[jj] = We[(long)jj * a.F];
    unsigned p0[4], p1[4], p2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) split3_pk(v[2 * m], v[2 * m + 1], p0[m], p1[m], p2[m]);
    unsigned short* O = a.wep + ((((long)net * (a.F / 32) + ft) * 4 + t) * 3) * 512 + lane * 8;
    *reinterpret_cast<u32x4*>(O) = (u32x4){p0[0], p0[1], p0[2], p0[3]};
    *reinterpret_cast<u32x4*>(O + 512) = (u32x4){p1[0], p1[1], p1[2], p1[3]};
    *reinterpret_cast<u32x4*>(O + 1024) = (u32x4){p2[0], p2[1], p2[2], p2[3]};
}

struct IqnEmbed3Args {
    const unsigned short* cosp;  // [V * N][4][3][64][8]
    const unsigned short* wep;   // [n_packed][F / 32][4][3][64][8]
    const float* const* wbase;   // [V] (bias)
    const float* psi;
    float* x;
    long be_off;
    int K, N, F, n_packed;       // virtual net v reads packed net v < n_packed ? v : v - K  (the two target sets share one)
};
// grid (f tiles / 4, virtual net, fraction group): a wave keeps its tile's We fragments (48 registers), psi tile and bias for the
// fractions of its group.  The cos fragments of a fraction are copied ONCE per workgroup into LDS by LDS-DMA (the four waves of a
// workgroup hold four feature tiles of the same (virtual net, fraction group) and read identical cos fragments: the
// register version fetched them four times from L2, one step ahead, and waited for them -- 163 us where its 24 MFMAs per
// 4 KB tile would allow ~45 and the 476 MB of stores ~95).  Two 12 KB buffers; per fraction: counted vmcnt (this wave's
// three copies of the fraction have landed: only the 16 tile stores issued behind them may be outstanding), one barrier
// (everybody's copies have landed, nobody still reads the other buffer), the copies of the next fraction, 12 ds_read_b128.
__global__ __launch_bounds__(256) void k_iqn_embed3l(IqnEmbed3Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char e3_lds[];  // [2][12][1024]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
    const int ft = min((int)blockIdx.x * 4 + wave, a.F / 32 - 1);  // (a wave past the last tile repeats it: same barriers)
    const bool live = (int)blockIdx.x * 4 + wave < a.F / 32;
    const int v = blockIdx.y, type = v / a.K, k = v - type * a.K;
    const int nq = a.N / (int)gridDim.z, q0 = blockIdx.z * nq;
    const int f0 = ft * 32, pv = v < a.n_packed ? v : v - a.K;
    const unsigned short* Wf = a.wep + (((long)pv * (a.F / 32) + ft) * 12) * 512 + lane * 8;
    bf16x8 wf[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) wf[t][p] = *reinterpret_cast<const bf16x8*>(Wf + (t * 3 + p) * 512);
    const unsigned long cos0 = (unsigned long)(a.cosp + ((long)(v * a.N + q0) * 12) * 512);  // 12 KB per fraction
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&e3_lds[0];
    auto copy = [&](int q, int buf) {  // pieces wave, wave + 4, wave + 8 of fraction q
#pragma unroll
        for (int i = 0; i < 3; ++i)
            dma16(lane * 16, cos0 + (unsigned long)q * 12288 + (unsigned long)(wave + 4 * i) * 1024, lds0 + buf * 12288 + (wave + 4 * i) * 1024);
    };
    copy(0, 0);
    const float* P = a.wbase[v];
    const float* psi = a.psi + (long)((type == 0 ? 0 : a.K) + k) * a.F * 32;
    float be[16], ps[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = f0 + mfma_row(i, h);
        be[i] = P[a.be_off + f];
        ps[i] = psi[(long)f * 32 + r];
    }
    for (int q = 0; q < nq; ++q) {
        const int buf = q & 1;
        // this wave's copies of fraction q: issued before the 16 stores of fraction q - 1 (none for q = 0)
        if (q == 0 || !live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a wave without a tile issues no stores)
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (q + 1 < nq) copy(q + 1, buf ^ 1);
        const unsigned char* C = e3_lds + buf * 12288 + lane * 16;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 c0 = *LDS_PTR(const bf16x8, C + (t * 3 + 0) * 1024), c1 = *LDS_PTR(const bf16x8, C + (t * 3 + 1) * 1024),
                         c2 = *LDS_PTR(const bf16x8, C + (t * 3 + 2) * 1024);
            acc = mfma_bf16(wf[t][2], c0, acc);
            acc = mfma_bf16(wf[t][0], c2, acc);
            acc = mfma_bf16(wf[t][1], c1, acc);
            acc = mfma_bf16(wf[t][1], c0, acc);
            acc = mfma_bf16(wf[t][0], c1, acc);
            acc = mfma_bf16(wf[t][0], c0, acc);
        }
        float* X = a.x + (long)(v * a.N + q0 + q) * a.F * 32;
        if (live) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_nontemporal_store(fmaxf(acc[i] + be[i], 0.f) * ps[i], X + (long)(f0 + mfma_row(i, h)) * 32 + r);
        }
    }
}

// Z[slot][action][b] = b1[action] + the Dense_1 chunk partials in chunk order: one workgroup per (virtual net, fraction).
struct IqnZArgs {
    const float* qpart;         // [V * N][J / 32][32 (action)][32]
    const float* const* wbase;  // [V]
    float* z;                   // [V * N][A][32]
    long b1_off;
    int N, NJC, A;
};
__global__ __launch_bounds__(256) void k_iqn_z(IqnZArgs a) {
    const int slot = blockIdx.x;
    const float* P = a.wbase[slot / a.N];
    for (int e = threadIdx.x; e < a.A * 32; e += 256) {
        const float* p = a.qpart + (long)slot * a.NJC * 1024 + e;
        float s = 0.f;
        for (int c = 0; c < a.NJC; c += 4) {  // NJC is a multiple of 4 (J a multiple of 128); chunk order
            const float x0 = p[(c + 0) * 1024], x1 = p[(c + 1) * 1024], x2 = p[(c + 2) * 1024], x3 = p[(c + 3) * 1024];
            s = (((s + x0) + x1) + x2) + x3;
        }
        a.z[(long)slot * a.A * 32 + e] = s + P[a.b1_off + (e >> 5)];
    }
}

// Greedy target action, targets, quantile Huber loss and its gradient from the Z of the three virtual nets of head k.
// One workgroup per head: thread = (sample b = t & 31, group g = t >> 5).
struct IqnLossArgs {
    const float* z;  // [V * N][A][32]
    int K, N, A, B, Bdiv;
    const int32_t* action;
    const float* reward;
    const uint8_t* terminal;
    const float* tau;  // [K][3][N][B]
    float gamma_n;
    float* dq;      // [K][N][A][32]  dL/dZ_online
    float* losses;  // [K]
    int32_t* count;
    double* cum;
    int finish_step;
    float* dbg;  // [K][(2 N + 32 + 1)][32]: Z_online(a) rows, Z_target(a*) rows, q_select rows (32), a* row -- tests
    const unsigned* gate_err;  // nonzero: a bounded wait of an earlier step's fused Dense_0 update (iqn_gemm.h, IqnD0Gate) gave up -> NaN losses
};
__global__ __launch_bounds__(256) void k_iqn_loss(IqnLossArgs a) {
    extern __shared__ float sm[];
    const int k = blockIdx.x, t = threadIdx.x, b = t & 31, g = t >> 5, N = a.N;
    float* zon = sm;             // [N][32]
    float* zval = zon + N * 32;  // [N][32]
    float* qsel = zval + N * 32; // [32][32]
    float* red = qsel + 32 * 32; // [8][32]
    __shared__ int astar[32], act[32];
    auto zat = [&](int v, int q, int ac) { return a.z[((long)(v * N + q) * a.A + ac) * 32 + b]; };
    if (t < 32) act[t] = t < a.B ? a.action[t] : 0;
    // mean over the selection fractions of the target net's Z, per action (fraction order)
    for (int ac = g; ac < a.A; ac += 8) {
        float s = 0.f;
        for (int q0 = 0; q0 < N; q0 += 8) {
            float zz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) zz[u] = q0 + u < N ? zat(a.K + k, q0 + u, ac) : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += zz[u];
        }
        qsel[ac * 32 + b] = s / (float)N;
    }
    __syncthreads();
    if (t < 32) {
        int best = 0;
        float bv = qsel[t];
        for (int ac = 1; ac < a.A; ++ac) {
            const float x = qsel[ac * 32 + t];
            if (x > bv) { bv = x; best = ac; }
        }
        astar[t] = best;
    }
    __syncthreads();
    for (int q = g; q < N; q += 8) {
        zval[q * 32 + b] = zat(2 * a.K + k, q, astar[b]);
        zon[q * 32 + b] = zat(k, q, act[b]);
    }
    __syncthreads();
    const bool live = b < a.B;
    const float rew = live ? a.reward[b] : 0.f;
    const float cont = (live && a.terminal[b]) ? 0.f : a.gamma_n;  // (1 - terminal) * gamma^n
    float lsum = 0.f;
    for (int j = g; j < N; j += 8) {
        const float z = zon[j * 32 + b];
        const float tj = live ? a.tau[(((long)k * 3 + 0) * N + j) * a.B + b] : 0.5f;
        float gsum = 0.f, ls = 0.f;
        for (int i = 0; i < N; ++i) {
            const float d = (rew + cont * zval[i * 32 + b]) - z;
            const float ad = fabsf(d);
            const float w = fabsf(tj - (d < 0.f ? 1.f : 0.f));
            ls += w * (ad <= 1.f ? 0.5f * d * d : ad - 0.5f);
            gsum += w * (ad <= 1.f ? d : (d > 0.f ? 1.f : -1.f));
        }
        lsum += ls;
        const float dz = live ? -gsum / ((float)a.Bdiv * (float)N) : 0.f;
        for (int ac = 0; ac < a.A; ++ac)
            a.dq[(((long)k * N + j) * a.A + ac) * 32 + b] = ac == act[b] ? dz : 0.f;
    }
    red[g * 32 + b] = live ? lsum / (float)N : 0.f;
    __syncthreads();
    if (a.dbg) {
        float* D = a.dbg + (long)k * (2 * N + 33) * 32;
        for (int e = t; e < N * 32; e += 256) { D[e] = zon[e]; D[N * 32 + e] = zval[e]; }
        for (int e = t; e < 32 * 32; e += 256) D[2 * N * 32 + e] = e < a.A * 32 ? qsel[e] : 0.f;
        if (t < 32) D[(2 * N + 32) * 32 + t] = (float)astar[t];
    }
    if (t == 0) {
        float s = 0.f;
        for (int x = 0; x < 32; ++x) {  // sample order, then group order
            float sb = 0.f;
            for (int gg = 0; gg < 8; ++gg) sb += red[gg * 32 + x];
            s += sb;
        }
        const float loss = s / (float)a.Bdiv;
        a.losses[k] = (a.gate_err && *a.gate_err) ? __uint_as_float(0x7fc00000u) : loss;
        if (a.finish_step) {
            a.count[k] += 1;
            a.cum[k] = a.cum[k] + (double)loss;
        }
    }
}

// dL/dh of every fraction block + Dense_1 / Dense_0-bias gradients.  grid = (J / 32 chunks, head).
struct IqnDhArgs {
    const float* hbuf;  // [V * N][J][32]
    const float* dq;    // [K][N][A][32]
    const float* const* wbase;
    long w1_off, gP, g_b0_off, g_w1_off, g_b1_off;
    int K, N, J, A;
    float* dh;    // [K][N][J][32]
    float* hpart; // [QG][K][J * A + J + A]: Dense_1 kernel, Dense_0 bias, Dense_1 bias gradient sums over the fractions of group qg
};
// grid = (J / 32 chunks, head, fraction group): 80 workgroups walking all N blocks in turn were one latency chain per block
// (90 us at N = 32); the groups' gradient partials are added in group order by k_iqn_head_grad_sum.
__global__ __launch_bounds__(256) void k_iqn_dh(IqnDhArgs a) {
    __shared__ float hs[32][33], dqs[32 * 32], w1s[32 * 32];
    const int jc = blockIdx.x, k = blockIdx.y, qg = blockIdx.z, t = threadIdx.x, b = t & 31, jj = t >> 5;
    const int nq = a.N / (int)gridDim.z, q_begin = qg * nq;
    const float* w1 = a.wbase[k] + a.w1_off + (long)jc * 32 * a.A;
    for (int e = t; e < 32 * a.A; e += 256) w1s[e] = w1[e];
    float gw[4] = {0.f, 0.f, 0.f, 0.f}, gb0[4] = {0.f, 0.f, 0.f, 0.f}, gb1 = 0.f;
    for (int q = q_begin; q < q_begin + nq; ++q) {
        const float* hb = a.hbuf + ((long)(k * a.N + q) * a.J + jc * 32) * 32;
        const float* dq = a.dq + ((long)k * a.N + q) * a.A * 32;
        __syncthreads();  // the previous block's tiles have been consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) hs[jj + 8 * i][b] = hb[(jj + 8 * i) * 32 + b];
        for (int e = t; e < a.A * 32; e += 256) dqs[e] = dq[e];
        __syncthreads();
        float* dh = a.dh + ((long)(k * a.N + q) * a.J + jc * 32) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int jl = jj + 8 * i;
            float d = 0.f;
            for (int ac = 0; ac < a.A; ++ac) d = fmaf(w1s[jl * a.A + ac], dqs[ac * 32 + b], d);
            d = hs[jl][b] > 0.f ? d : 0.f;
            dh[jl * 32 + b] = d;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) d += __shfl_xor(d, o);
            gb0[i] += d;
        }
        // Dense_1 weight gradient: thread -> elements (jl, ac) = (o / A, o % A), o = t + 256 m; sum over the samples
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int o = t + 256 * m;
            if (o < 32 * a.A) {
                const int jl = o / a.A, ac = o - jl * a.A;
                float s = 0.f;
#pragma unroll
                for (int x = 0; x < 32; ++x) s = fmaf(hs[jl][x], dqs[ac * 32 + x], s);
                gw[m] += s;
            }
        }
        if (jc == 0 && t < a.A) {
            float s = 0.f;
#pragma unroll
            for (int x = 0; x < 32; ++x) s += dqs[t * 32 + x];
            gb1 += s;
        }
    }
    const long hn = (long)a.J * a.A + a.J + a.A;
    float* G = a.hpart + ((long)qg * a.K + k) * hn;  // [J * A | J | A]
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (b == 0) G[(long)a.J * a.A + jc * 32 + jj + 8 * i] = gb0[i];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int o = t + 256 * m;
        if (o < 32 * a.A) G[(long)jc * 32 * a.A + o] = gw[m];
    }
    if (jc == 0 && t < a.A) G[(long)a.J * a.A + a.J + t] = gb1;
}

struct IqnHeadGradSumArgs {
    const float* hpart;  // [QG][K][J * A + J + A]
    float* grad;
    long gP, g_b0_off, g_w1_off, g_b1_off;
    int K, J, A, QG;
};
__global__ __launch_bounds__(256) void k_iqn_head_grad_sum(IqnHeadGradSumArgs a) {
    const int k = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x, wn = (long)a.J * a.A, hn = wn + a.J + a.A;
    if (e >= hn) return;
    float s = a.hpart[(long)k * hn + e];
    for (int g = 1; g < a.QG; ++g) s += a.hpart[((long)g * a.K + k) * hn + e];
    float* G = a.grad + (long)k * a.gP;
    G[e < wn ? a.g_w1_off + e : (e < wn + a.J ? a.g_b0_off + (e - wn) : a.g_b1_off + (e - wn - a.J))] = s;
}

// Backward of the Hadamard product and of the embedding, one wave per (head, 32-feature tile), fractions in order:
//   e, phi recomputed (the forward's MFMA);  dpsi[f][b] += dx * phi;  dphi = dx * psi * [e > 0];
//   dWe[i][f] += sum_b cos[i][b] dphi[f][b]  (MFMA over the 32 samples, dphi through a per-wave LDS tile);  dbe[f] += sum_b dphi.
// The same backward on the bf16 matrix cores: the recomputed embedding from the pre-split planes (k_iqn_we_pack, cosp:
// 24 products instead of 32 f32 MFMAs of twice the length), dL/dWe from the A-fragment planes of cos (cosa) and dphi split
// in the kernel after its trip through the per-wave LDS tile (8 split3_pk per lane and fraction): 24 products instead of
// 32.  grid = (f tiles / 4, head, fraction group): the N fractions of a tile are dealt to gridDim.z workgroups.
struct IqnEmbedBwd3Args {
    const unsigned short* cosp;  // [V * N][12][512]
    const unsigned short* cosa;  // [V * N][12][512]
    const unsigned short* wep;   // [n_packed][F / 32][12][512]  (online nets first)
    const float* const* wbase;
    const float* psi;
    const float* dx;
    float* dpsi;
    float* gpart;
    long be_off;
    int K, N, F;
};
__global__ __launch_bounds__(256, 2) void k_iqn_embed_bwd3(IqnEmbedBwd3Args a) {
    __shared__ float tile[4][32][33];
    extern __shared__ __attribute__((aligned(1024))) unsigned short eb3_cs[];  // [buffer 2][cosp | cosa][12 * 512]: 48 KB (dynamic)
    unsigned short (*cs)[2][12 * 512] = reinterpret_cast<unsigned short (*)[2][12 * 512]>(eb3_cs);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
    const int k = blockIdx.y, qg = blockIdx.z;
    const int nq = a.N / (int)gridDim.z, q_begin = qg * nq;
    const bool live = (int)blockIdx.x * 4 + wave < a.F / 32;
    const int ft = min((int)blockIdx.x * 4 + wave, a.F / 32 - 1);
    const int f0 = ft * 32;
    const float* P = a.wbase[k];
    bf16x8 wf[4][3];
    {
        const unsigned short* Wf = a.wep + (((long)k * (a.F / 32) + ft) * 12) * 512 + lane * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) wf[t][p] = *reinterpret_cast<const bf16x8*>(Wf + (t * 3 + p) * 512);
    }
    float be[16], ps[16], dps[16], dbe[16];
    const float* psi = a.psi + (long)k * a.F * 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int f = f0 + mfma_row(i, h);
        be[i] = P[a.be_off + f];
        ps[i] = psi[(long)f * 32 + r];
        dps[i] = 0.f;
        dbe[i] = 0.f;
    }
    f32x16 gw0, gw1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { gw0[i] = 0.f; gw1[i] = 0.f; }
    const unsigned lds_cs = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)&cs[0][0][0];
    // this wave's quarter of the 24 KB [cosp | cosa] of fraction q -> buffer `buf` (six 1 KB pieces)
    auto stage = [&](int q, int buf) {
        const long slot = (long)k * a.N + q;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int piece = wave * 6 + c, which = piece >= 12, off = (piece - 12 * which) * 512;  // shorts
            const unsigned short* src = (which ? a.cosa : a.cosp) + slot * (12 * 512) + off;
            dma16((unsigned)lane * 16, (unsigned long)src, lds_cs + (unsigned)(((buf * 2 + which) * 12 * 512 + off) * 2));
        }
    };
    float dxr[2][16];
    auto fetch_dx = [&](int q, int st) {
        const float* DX = a.dx + ((long)k * a.N + q) * a.F * 32;
#pragma unroll
        for (int i = 0; i < 16; ++i) dxr[st][i] = DX[(long)(f0 + mfma_row(i, h)) * 32 + r];
    };
    stage(q_begin, 0);
    fetch_dx(q_begin, 0);
#define EB3_STEP(qi, st)                                                                                 \
    {                                                                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                    \
        if ((qi) + 1 < nq) {                                                                             \
            stage(q_begin + (qi) + 1, (st) ^ 1);                                                         \
            fetch_dx(q_begin + (qi) + 1, (st) ^ 1);                                                      \
        }                                                                                                \
        const unsigned short* Cb = &cs[st][0][0] + lane * 8;                                             \
        const unsigned short* Ca = &cs[st][1][0] + lane * 8;                                             \
        f32x16 acc;                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[i] = 0.f;                                     \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                  \
            const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(Cb + (t * 3 + 0) * 512);                  \
            const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(Cb + (t * 3 + 1) * 512);                  \
            const bf16x8 c2 = *reinterpret_cast<const bf16x8*>(Cb + (t * 3 + 2) * 512);                  \
            acc = mfma_bf16(wf[t][2], c0, acc);                                                          \
            acc = mfma_bf16(wf[t][0], c2, acc);                                                          \
            acc = mfma_bf16(wf[t][1], c1, acc);                                                          \
            acc = mfma_bf16(wf[t][1], c0, acc);                                                          \
            acc = mfma_bf16(wf[t][0], c1, acc);                                                          \
            acc = mfma_bf16(wf[t][0], c0, acc);                                                          \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                 \
            const int fl = mfma_row(i, h);                                                               \
            const float e = acc[i] + be[i];                                                              \
            const float dx = dxr[st][i];                                                                 \
            dps[i] = fmaf(dx, fmaxf(e, 0.f), dps[i]);                                                    \
            const float dphi = e > 0.f ? dx * ps[i] : 0.f;                                               \
            dbe[i] += dphi;                                                                              \
            tile[wave][fl][r] = dphi;                                                                    \
        }                                                                                                \
        __builtin_amdgcn_wave_barrier();                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        /* dWe[i][f] += sum_b cos[i][b] dphi[f][b]: A = cosa fragments (row i, k = sample), B = dphi (k = sample 16 t + 8 h + jj, col f = r) */ \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                  \
            float dv[8];                                                                                 \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) dv[jj] = tile[wave][r][16 * t + 8 * h + jj]; \
            unsigned p0[4], p1[4], p2[4];                                                                \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) split3_pk(dv[2 * m], dv[2 * m + 1], p0[m], p1[m], p2[m]); \
            const bf16x8 d0 = planes8(p0), d1 = planes8(p1), d2 = planes8(p2);                           \
            _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                           \
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Ca + ((rt * 2 + t) * 3 + 0) * 512);   \
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Ca + ((rt * 2 + t) * 3 + 1) * 512);   \
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(Ca + ((rt * 2 + t) * 3 + 2) * 512);   \
                f32x16& g = rt ? gw1 : gw0;                                                              \
                g = mfma_bf16(a2, d0, g);                                                                \
                g = mfma_bf16(a0, d2, g);                                                                \
                g = mfma_bf16(a1, d1, g);                                                                \
                g = mfma_bf16(a1, d0, g);                                                                \
                g = mfma_bf16(a0, d1, g);                                                                \
                g = mfma_bf16(a0, d0, g);                                                                \
            }                                                                                            \
        }                                                                                                \
        __builtin_amdgcn_wave_barrier();                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
    }
    for (int qi = 0; qi < nq; qi += 2) {
        EB3_STEP(qi, 0)
        if (qi + 1 < nq) EB3_STEP(qi + 1, 1)
    }
#undef EB3_STEP
    if (!live) return;
    float* DP = a.dpsi + ((long)qg * a.K + k) * a.F * 32;
    float* G = a.gpart + ((long)qg * a.K + k) * 65 * a.F;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int fl = mfma_row(i, h);
        DP[(long)(f0 + fl) * 32 + r] = dps[i];
        float d = dbe[i];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) d += __shfl_xor(d, o);
        if (r == 0) G[64L * a.F + f0 + fl] = d;
        G[(long)mfma_row(i, h) * a.F + f0 + r] = gw0[i];
        G[(long)(32 + mfma_row(i, h)) * a.F + f0 + r] = gw1[i];
    }
}

// dL/dWe, dL/dbe = the fraction groups' partials added in group order -> gradient arena.  One float4 per thread.
struct IqnEmbedGradSumArgs {
    const float* gpart;  // [QG][K][65][F]
    float* grad;
    long gP, g_we_off, g_be_off;
    int K, F, QG;
};
__global__ __launch_bounds__(256) void k_iqn_embed_grad_sum(IqnEmbedGradSumArgs a) {
    const int k = blockIdx.y;
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4, n = 65L * a.F;
    if (e >= n) return;
    const float* p = a.gpart + (long)k * n + e;
    float4 s = *reinterpret_cast<const float4*>(p);
    for (int g = 1; g < a.QG; ++g) {
        const float4 y = *reinterpret_cast<const float4*>(p + (long)g * a.K * n);
        s.x += y.x; s.y += y.y; s.z += y.z; s.w += y.w;
    }
    float* G = a.grad + (long)k * a.gP;
    const long we_n = 64L * a.F;  // F is a multiple of 4: a float4 never straddles the two leaves
    *reinterpret_cast<float4*>(e < we_n ? G + a.g_we_off + e : G + a.g_be_off + (e - we_n)) = s;
}

// mean over the N fractions of Z (acting): q[a] of ONE state = lane 0 of every block.  One wave per action.
struct IqnQOutArgs {
    const float* qpart;  // [N][J / 32][32][32]
    const float* params;
    long b1_off;
    int N, NJC, A, n;
    float* q_out;      // [n][A]
    int32_t* action;   // [n] or nullptr
};
__global__ __launch_bounds__(256) void k_iqn_q_out(IqnQOutArgs a) {
    __shared__ float qs[32 * 32];
    for (int e = threadIdx.x; e < a.A * 32; e += 256) {
        const int ac = e >> 5, b = e & 31;
        float s = 0.f;
        for (int q = 0; q < a.N; ++q) {
            float z = 0.f;
            for (int c = 0; c < a.NJC; ++c) z += a.qpart[(((long)q * a.NJC + c) * 32 + ac) * 32 + b];
            s += z + a.params[a.b1_off + ac];
        }
        s /= (float)a.N;
        qs[e] = s;
        if (b < a.n) a.q_out[b * a.A + ac] = s;
    }
    __syncthreads();
    if (a.action && (int)threadIdx.x < a.n) {
        int best = 0;
        float bv = qs[threadIdx.x];
        for (int ac = 1; ac < a.A; ++ac)
            if (qs[ac * 32 + threadIdx.x] > bv) { bv = qs[ac * 32 + threadIdx.x]; best = ac; }
        a.action[threadIdx.x] = best;
    }
}
