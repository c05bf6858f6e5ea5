// MLP ("fc") i-DQN gradient step, ONE launch per step: every global read requested at kernel entry, both forwards side by side,
// the gradient assembled in LDS, Adam in one coalesced pass.
//
// Reference: slimdqn/networks/architectures/dqn.py:61-70 (squeeze, [Dense + ReLU] x len(features), Dense), idqn.py:96-124
// (TD target, loss, value_and_grad, optax.adam).  An 11 k-parameter head is < 1 us of matrix work: the step is the length of
// its chain of dependent phases and of the memory round trips inside them.  k_fc_step_mfma (fc_kernels.h) runs one workgroup
// per head through
//   zero LDS | inputs | 3 x (stage W_target to LDS | layer) | 3 x (stage W_online | layer) | TD | 3 x (restage | gradients)
// and a second launch applies Adam: 58 us at K = 3, [100, 100] (profiles/r5_final_loop.txt).  A first one-launch version
// (forwards side by side, weight operands from global in 16-step chunks, Adam operands fetched per gradient tile) took 40 us:
// phase stamps (profiles/r6_fc_phases.txt) showed every phase waiting for global loads it had only just issued.  Here:
//     A second version that kept the forward's weight operands in registers (one dword load per lane and MFMA step) was no
//     faster: 1,000 dword load instructions per workgroup are a longer queue than their latency.
//   * kernel entry requests what the step reads with 16-byte loads in arena order: both nets' parameters (6 + 6 loads per
//     thread), the minibatch, the rewards / actions / terminals; the online theta stays in registers until the last phase
//     (m and v follow behind the forward);
//   * both nets' matrices go to LDS out of those registers, row-major with an odd pitch: the forward tiles read them
//     column-wise, the data gradients row-wise, both conflict-free; the target net's copy is dead after its forward and its
//     LDS becomes the gradient arena;
//   * the target net (waves 4-7) and the online net (waves 0-3) run their forwards at the same time, one column tile per wave,
//     one barrier per layer;
//   * a layer's data-gradient and weight-gradient tiles are dealt to the eight waves together; every gradient tile lands in an
//     LDS copy of the head's arena, and ONE pass writes it out coalesced and applies optax.adam to the registers from entry.
// Batches of <= 32 samples, layer widths <= 128, heads of <= 12 k parameters; anything else runs k_fc_step_mfma / k_fc_step_lds.
// Exact f32 products (v_mfma_f32_32x32x2_f32), k-ordered sums.
#pragma once
#include "dense0_update.h"
#include "fc_kernels.h"

#define FCP_NPT 24   // parameters per thread of the flat (Adam) view: heads of up to 512 x 24 = 12,288 arena floats
#define FCP_LDS_BUDGET (160 * 1024 - 256)

struct FcParPlan {
    long floats;                  // LDS floats (0: the net does not fit this kernel)
    int drows;                    // rows of a target / delta buffer
    long act_row[FC_MAX_LAYERS + 1];  // first row of the online activations of layer input l
    long buf_off, wo_off, wt_off, misc_off;  // wo / wt: the two nets' parameters in ARENA order; wt later holds the gradient
};
static inline FcParPlan fc_par_plan(const FcNet& n, long P) {
    FcParPlan p;
    memset(&p, 0, sizeof(p));
    long rows = 0;
    bool ok = P <= 512L * FCP_NPT && P % 4 == 0;
    for (int l = 0; l <= n.L; ++l) { p.act_row[l] = rows; rows += n.d[l]; }
    for (int l = 0; l < n.L; ++l) ok = ok && n.d[l] <= 128 && n.d[l + 1] <= 128;  // one column tile per wave of a group of four
    ok = ok && 32L * n.d[0] <= 8L * FCM_T;
    p.drows = (n.dmax + 31) / 32 * 32;
    p.buf_off = rows * FCM_BSP;
    long off = p.buf_off + 2L * p.drows * FCM_BSP + 32L * FCM_BSP;  // (32 rows of slack: edge tiles read rows past a buffer's end)
    off = (off + 3) & ~3L;
    p.wo_off = off;
    off += P + 32;        // (+ 32: the last matrix's edge tile reads up to 31 floats past the arena's end)
    p.wt_off = off;
    off += P + 32;
    p.misc_off = off;
    p.floats = off + 128;
    if (!ok || p.floats * 4 > FCP_LDS_BUDGET) p.floats = 0;
    return p;
}

__global__ __launch_bounds__(FCM_T) void k_fc_step_par(FcArgs a, FcParPlan p, AdamConsts ad, float* theta, float* mu, float* nu, int do_adam, int prof) {
    extern __shared__ __attribute__((aligned(16))) float fl[];
    const int k = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, bl = lane & 31, h = lane >> 5;
    const int half = wave >> 2, wsub = wave & 3;  // waves 0-3: the online net, waves 4-7: the target net
    const FcNet& n = a.net;
    const int B = a.B, A = n.d[n.L];
    const long P = a.P;
    auto act = [&](int l) { return fl + p.act_row[l] * FCM_BSP; };
    float* dA = fl + p.buf_off;                       // target ping, then delta buffers
    float* dB = dA + (long)p.drows * FCM_BSP;
    float* Gs = fl + p.wt_off;                        // (behind the forwards)
    float* misc = fl + p.misc_off;
    float *sq = misc + 32, *bc = misc + 64;
    const float* po = a.online + (long)k * P;
    const float* pt = a.target + (long)k * P;
    float* G = a.grad + (long)k * P;
    float* TH = theta + (long)k * P;
    float* MU = mu + (long)k * P;
    float* NU = nu + (long)k * P;
    // debugging (IDQN_FC_PROF, debug build): shader-clock stamps of thread 0 at the phase boundaries -> a.ws
#define FCP_STAMP(i) if (prof && t == 0) reinterpret_cast<long long*>(a.ws)[k * 16 + (i)] = clock64();
    FCP_STAMP(0)
    // ---- everything the step reads from global memory is requested HERE, before anything waits, in 16-byte pieces
    constexpr int NV = FCP_NPT / 4;
    float4 th4[NV], tt4[NV], mm4[NV], vv4[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const long e = 4L * (t + FCM_T * j);
        th4[j] = e < P ? *reinterpret_cast<const float4*>(TH + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        tt4[j] = e < P ? *reinterpret_cast<const float4*>(pt + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float xin[8], xin2[8];  // the minibatch: element e = t + 512 j of [32][d0] (d0 <= 128)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = t + FCM_T * j;
        const bool in = e < B * n.d[0];
        xin[j] = in ? a.s[e] : 0.f;
        xin2[j] = in ? a.s2[e] : 0.f;
    }
    float r_b = 0.f;
    int a_b = 0, t_b = 1;
    if (t < 32 && t < B) { r_b = a.reward[t]; a_b = a.action[t]; t_b = (int)a.terminal[t]; }
    const float w_b = (t < 32 && t < B && a.is_weight) ? a.is_weight[t] : 1.0f;
    // nothing but finite numbers ever lives in this LDS (edge tiles multiply junk rows by zeros / their results are dropped)
    for (long e = t; e < p.wo_off / 4; e += FCM_T) reinterpret_cast<float4*>(fl)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < 32) { fl[p.wo_off + P + t] = 0.f; fl[p.wt_off + P + t] = 0.f; }
    if (t < 128) misc[t] = 0.f;
    __syncthreads();
    FCP_STAMP(1)
    // ---- inputs, transposed (rows past the batch end are zero inputs; they carry no loss weight)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = t + FCM_T * j;
        if (e < 32 * n.d[0]) {
            const int b = e / n.d[0], i = e - b * n.d[0];
            dA[i * FCM_BSP + b] = xin2[j];
            fl[i * FCM_BSP + b] = xin[j];
        }
    }
    // ---- both nets' parameters to LDS as they are (arena order, 16 bytes per lane): a forward tile reads a matrix along its rows
    // (conflict-free for any pitch), a data-gradient tile down its columns (pitch dout: a few-way conflict on a read that is far
    // from the bottleneck) -- no index arithmetic, no padding to maintain
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const long e = 4L * (t + FCM_T * j);
        if (e < P) {
            *reinterpret_cast<float4*>(fl + p.wo_off + e) = th4[j];
            *reinterpret_cast<float4*>(fl + p.wt_off + e) = tt4[j];
        }
    }
    __syncthreads();
    FCP_STAMP(2)
    const float* tq = nullptr;  // the target net's Q
    float g_b = 0.f;
    // ---- forwards: one column tile per wave and layer; both operands of an MFMA step are one conflict-free ds_read_b32 per lane
    {
        const float* Wn = fl + (half == 0 ? p.wo_off : p.wt_off);
        float *cur = dA, *nxt = dB;
        for (int l = 0; l < n.L; ++l) {
            const int din = n.d[l], dout = n.d[l + 1], ks = (din + 1) / 2, ldw = dout;
            const bool relu = l != n.L - 1;
            const float* inT = half == 0 ? act(l) : cur;
            float* outT = half == 0 ? act(l + 1) : nxt;
            // column tile ct on wave (ct + half) % 4 of the group: a one-tile layer (the Q head) then runs its two nets on two
            // different SIMDs instead of queueing both chains on SIMD 0
            const int ct = (wsub - half) & 3;
            if (ct * 32 < dout) {
                const int col = ct * 32 + bl;
                const float bv = Wn[n.b_off[l] + min(col, dout - 1)];
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* Ap = inT + h * FCM_BSP + bl;             // A[b = bl][k = 2 s + h]
                const float* Bp = Wn + n.w_off[l] + h * ldw + col;    // B[k = 2 s + h][col] (steps past din masked)
                if (din & 1) {  // (an odd input width: the second half-wave's last step has no row)
                    for (int s0 = 0; s0 < ks; ++s0) acc = mfma32(Ap[2 * s0 * FCM_BSP], 2 * s0 + h < din ? Bp[2 * s0 * ldw] : 0.f, acc);
                } else {
#pragma unroll 2
                    for (int s0 = 0; s0 < ks; ++s0) acc = mfma32(Ap[2 * s0 * FCM_BSP], Bp[2 * s0 * ldw], acc);
                }
                if (col < dout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[r] + bv;
                        outT[col * FCM_BSP + mfma_row(r, h)] = relu ? fmaxf(v, 0.f) : v;
                    }
                }
            }
            if (l == n.L - 1 && t == FCM_T - 1) {
                // reciprocal Adam bias corrections of this step (optax: t = count + 1; two double pows): on a wave that has no
                // tile of the Q head's layer, instead of holding the first barrier back
                const double tt = (double)(a.count[k] + 1);
                const float r1 = 1.0f / (1.0f - (float)pow((double)a.adam_b1, tt)), r2 = 1.0f / (1.0f - (float)pow((double)a.adam_b2, tt));
                bc[0] = r1; bc[1] = r2;
                a.bcinv[2 * k] = r1; a.bcinv[2 * k + 1] = r2;
            }
            float* tmp = cur; cur = nxt; nxt = tmp;
            __syncthreads();
            FCP_STAMP(3 + l)
        }
        // cur = the target net's Q [A][BSP]
        const float* q = act(n.L);  // the online net's
        for (int e = t; e < B * A; e += FCM_T) {
            a.q_dbg[((long)(a.K + k) * B) * A + e] = cur[(e % A) * FCM_BSP + e / A];
            a.q_dbg[((long)k * B) * A + e] = q[(e % A) * FCM_BSP + e / A];
        }
        // ---- TD error, loss, dL/dq  (idqn.py:111-124): max over actions in action order
        tq = cur;
        if (t < 32) {
            const int b = t;
            float m = -INFINITY;
            for (int ac = 0; ac < A; ++ac) m = fmaxf(m, cur[ac * FCM_BSP + b]);
            float sqv = 0.f, g = 0.f;
            if (b < B) {
                const float tgt = r_b + (float)(1 - t_b) * a.gamma_n * m;
                const float td = q[a_b * FCM_BSP + b] - tgt;
                if (a.td_abs) a.td_abs[(long)k * B + b] = fabsf(td);
                sqv = w_b * td * td;
                g = 2.0f * w_b * td / (float)a.Bdiv;
            }
            sq[b] = sqv;
            g_b = g;  // dL/dq of the taken action
        }
    }
    if (do_adam) {  // Adam's other operands: back long before the last phase
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const long e = 4L * (t + FCM_T * j);
            mm4[j] = e < P ? *reinterpret_cast<const float4*>(MU + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            vv4[j] = e < P ? *reinterpret_cast<const float4*>(NU + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // delta = dL/dq, transposed [A rows][32 samples], in the buffer that does NOT hold the target's Q (still being read): its rows
    // up to the next multiple of 32 zero (activations were there); the target net's LDS copy, dead now, becomes the gradient arena
    float* delta = tq == dA ? dB : dA;
    float* dprev = tq == dA ? dA : dB;
    for (int e = t; e < 32 * 32; e += FCM_T) delta[(e >> 5) * FCM_BSP + (e & 31)] = 0.f;
    __syncthreads();  // (every wave is done with the target net's matrices and with delta's old contents)
    for (long e = t; e < P / 4; e += FCM_T) reinterpret_cast<float4*>(Gs)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < 32 && t < B) delta[a_b * FCM_BSP + t] = g_b;
    __syncthreads();
    FCP_STAMP(8)
    float loss_sum = 0.f;
    if (t == 0)
        for (int b = 0; b < 32; ++b) loss_sum += sq[b];
    // ---- backward, top down; per layer the data-gradient tiles first, then the weight-gradient tiles, round-robin over the waves;
    // every gradient lands in the LDS copy of the arena
    for (int l = n.L - 1; l >= 0; --l) {
        const int din = n.d[l], dout = n.d[l + 1];
        const float* inT = act(l);
        const int nti = (din + 31) / 32, nto = (dout + 31) / 32, nd = l > 0 ? nti : 0;
        // tasks round-robin over the waves: waves w and w + 4 share a SIMD, so with the data-gradient tiles (the long chains) on
        // waves 0-3 every SIMD ends up with the same number of MFMAs (a split by per-wave load left SIMDs idle: +4 k cycles)
        for (int task = wave; task < nd + nti * nto; task += FCM_T / 64) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (task < nd) {
                // dprev[i][b] = relu'(in[i][b]) * sum_o delta[o][b] * W[i][o]: 32 columns i, k = o in pairs
                const int ti = task, ks = (dout + 1) / 2, ldw = dout;
                const float* Ap = delta + h * FCM_BSP + bl;                                                          // A[b = bl][k = o = 2 s + h]
                const float* Bp = fl + p.wo_off + n.w_off[l] + (long)min(ti * 32 + bl, din - 1) * ldw + h;            // B[k = o][i = bl] (steps past dout masked)
                if (dout & 1) {
                    for (int s0 = 0; s0 < ks; ++s0) acc = mfma32(Ap[2 * s0 * FCM_BSP], 2 * s0 + h < dout ? Bp[2 * s0] : 0.f, acc);
                } else {
#pragma unroll 2
                    for (int s0 = 0; s0 < ks; ++s0) acc = mfma32(Ap[2 * s0 * FCM_BSP], Bp[2 * s0], acc);
                }
                const int i = ti * 32 + bl;
                if (i < din) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int b = mfma_row(r, h);
                        dprev[i * FCM_BSP + b] = inT[i * FCM_BSP + b] > 0.f ? acc[r] : 0.f;
                    }
                }
            } else {
                // gW[i][o] = sum_b inT[i][b] * delta[o][b]: 32 rows i x 32 columns o, k = the 32 samples
                const int tile = task - nd, ti = tile / nto, to = tile - ti * nto;
                const int o = to * 32 + bl;
                const float* Ap = inT + (long)(ti * 32 + bl) * FCM_BSP + h;    // A[i = bl][k = b = 2 s + h]
                const float* Bp = delta + (long)(to * 32 + bl) * FCM_BSP + h;  // B[k = b][o = bl]
                float av[16], bv[16];
#pragma unroll
                for (int s0 = 0; s0 < 16; ++s0) { av[s0] = Ap[2 * s0]; bv[s0] = Bp[2 * s0]; }
#pragma unroll
                for (int s0 = 0; s0 < 16; ++s0) acc = mfma32(av[s0], bv[s0], acc);
                if (o < dout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = ti * 32 + mfma_row(r, h);
                        if (i < din) Gs[n.w_off[l] + (long)i * dout + o] = acc[r];
                    }
                }
            }
        }
        for (int o = FCM_T - 1 - t; o < dout; o += FCM_T) {  // bias gradient, in sample order (the last waves: the shortest task lists)
            float s = 0.f;
            for (int b = 0; b < 32; ++b) s += delta[o * FCM_BSP + b];
            Gs[n.b_off[l] + o] = s;
        }
        __syncthreads();
        FCP_STAMP(9 + (n.L - 1 - l))
        float* tmp = delta; delta = dprev; dprev = tmp;
    }
    // ---- the gradient leaves LDS in arena order, 16 bytes per lane; optax.adam on the operands requested at kernel entry
    {
        const float rbc1 = bc[0], rbc2 = bc[1];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const long e = 4L * (t + FCM_T * j);
            if (e < P) {
                const float4 g = *reinterpret_cast<const float4*>(Gs + e);
                *reinterpret_cast<float4*>(G + e) = g;
                if (do_adam) {
                    adam_elem(ad, rbc1, rbc2, g.x, th4[j].x, mm4[j].x, vv4[j].x);
                    adam_elem(ad, rbc1, rbc2, g.y, th4[j].y, mm4[j].y, vv4[j].y);
                    adam_elem(ad, rbc1, rbc2, g.z, th4[j].z, mm4[j].z, vv4[j].z);
                    adam_elem(ad, rbc1, rbc2, g.w, th4[j].w, mm4[j].w, vv4[j].w);
                    *reinterpret_cast<float4*>(TH + e) = th4[j];
                    *reinterpret_cast<float4*>(MU + e) = mm4[j];
                    *reinterpret_cast<float4*>(NU + e) = vv4[j];
                }
            }
        }
    }
    FCP_STAMP(15)
    if (t == 0) {
        a.losses[k] = loss_sum / (float)a.Bdiv;
        if (a.finish_step) {
            a.count[k] += 1;
            a.cum[k] = a.cum[k] + (double)(loss_sum / (float)a.Bdiv);
        }
    }
}
