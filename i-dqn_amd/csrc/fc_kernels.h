// gfx950 kernel of the MLP ("fc") i-DQN gradient step -- LunarLander-sized nets (8 -> 100 -> 100 -> 4).
//
// Reference: slimdqn/networks/architectures/dqn.py:61-70 (squeeze, [Dense + ReLU] x len(features), Dense)
// and slimdqn/networks/idqn.py:105-124.  These nets are ~11 k parameters per head: the step is
// launch-latency bound, so ONE workgroup per head runs the whole forward (online on s, target on s'),
// the TD target with the wavefront max over actions, the loss and the backward in a single launch;
// gradients go to the arena and the shared Adam kernel applies them.  Plain f32 FMAs: the layer
// widths (8, 100, 4) are not MFMA-shaped and the total is ~1 MFLOP per head.
#pragma once
#include "common.h"

#define FC_MAX_WIDTH 512
#define FC_MAX_LAYERS (IDQN_MAX_FEATURES + 1)

struct FcNet {
    int L;                      // number of Dense layers (hidden + output)
    int d[FC_MAX_LAYERS + 1];   // d[0] = input dim, d[l+1] = width of layer l
    long w_off[FC_MAX_LAYERS], b_off[FC_MAX_LAYERS];
    int dmax;
};

struct FcArgs {
    FcNet net;
    const float* online;  // [K][P]
    const float* target;  // [K][P]
    float* grad;          // [K][P]
    long P;
    const float* s;       // [B][d0]
    const float* s2;      // [B][d0]
    const int32_t* action;
    const float* reward;
    const uint8_t* terminal;
    float gamma_n;
    int B, Bdiv, K;
    float* ws;            // [K][ (L+1) * B * dmax (online acts) + 2 * B * dmax (target ping-pong / deltas) + 2*B ]
    float* losses;        // [K]
    float* q_dbg;         // [2K][B][A]
    int32_t* count;       // [K] optax step counter (pre-increment)
    float* bcinv;         // [K][2] out: reciprocal Adam bias corrections of this step
    float adam_b1, adam_b2;
    double* cum;          // [K] running f64 loss sum (idqn.py:72)
    int finish_step;      // 1: also count += 1, cum += loss here (the Adam kernel reads bcinv, not count)
    const float* is_weight;  // [B] per-sample loss weights (prioritized-replay extension) or nullptr
    float* td_abs;           // [K][B] out: |TD error| per head and sample, or nullptr
};

// out[b][o] = (relu?)(bias[o] + sum_i in[b][i] * W[i][o])
__device__ __forceinline__ void fc_layer(const float* in, int in_ld, const float* W, const float* bias, float* out,
                                         int out_ld, int B, int din, int dout, bool relu) {
    for (int e = threadIdx.x; e < B * dout; e += blockDim.x) {
        int b = e / dout, o = e - b * dout;
        float s = bias[o];
        const float* x = in + (long)b * in_ld;
        for (int i = 0; i < din; ++i) s = fmaf(x[i], W[(long)i * dout + o], s);
        out[(long)b * out_ld + o] = relu ? fmaxf(s, 0.f) : s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_fc_step(FcArgs a) {
    const int k = blockIdx.x, t = threadIdx.x;
    const FcNet& n = a.net;
    const int B = a.B, dm = n.dmax, A = n.d[n.L];
    float* ws = a.ws + (long)k * ((long)(n.L + 3) * B * dm + 2 * B);
    float* acts = ws;                               // [L+1][B][dm]: acts[0] = input, acts[l+1] = layer l output
    float* tA = ws + (long)(n.L + 1) * B * dm;      // target ping
    float* tB = tA + (long)B * dm;                  // target pong / delta buffers in the backward
    float* qmax = tB + (long)B * dm;                // [B]
    float* sq = qmax + B;                           // [B]
    const float* po = a.online + (long)k * a.P;
    const float* pt = a.target + (long)k * a.P;
    float* G = a.grad + (long)k * a.P;
    // ---- target net on s'
    for (int e = t; e < B * n.d[0]; e += 256) tA[(long)(e / n.d[0]) * dm + e % n.d[0]] = a.s2[e];
    __syncthreads();
    float *cur = tA, *nxt = tB;
    for (int l = 0; l < n.L; ++l) {
        fc_layer(cur, dm, pt + n.w_off[l], pt + n.b_off[l], nxt, dm, B, n.d[l], n.d[l + 1], l != n.L - 1);
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    // wavefront max over actions: lanes = (sample, half), each half folds every other action, one swap
    for (int b0 = 0; b0 < B; b0 += 128) {
        int b = b0 + (t >> 6) * 32 + (t & 31), hh = (t & 63) >> 5;
        float m = -INFINITY;
        if (b < B)
            for (int ac = hh; ac < A; ac += 2) m = fmaxf(m, cur[(long)b * dm + ac]);
        m = fmaxf(m, __shfl_xor(m, 32));
        if (b < B && hh == 0) qmax[b] = m;
    }
    for (int e = t; e < B * A; e += 256) a.q_dbg[((long)(a.K + k) * B) * A + e] = cur[(long)(e / A) * dm + e % A];
    __syncthreads();
    // ---- online net on s, activations kept
    for (int e = t; e < B * n.d[0]; e += 256) acts[(long)(e / n.d[0]) * dm + e % n.d[0]] = a.s[e];
    __syncthreads();
    for (int l = 0; l < n.L; ++l)
        fc_layer(acts + (long)l * B * dm, dm, po + n.w_off[l], po + n.b_off[l], acts + (long)(l + 1) * B * dm, dm, B,
                 n.d[l], n.d[l + 1], l != n.L - 1);
    const float* q = acts + (long)n.L * B * dm;
    for (int e = t; e < B * A; e += 256) a.q_dbg[((long)k * B) * A + e] = q[(long)(e / A) * dm + e % A];
    // ---- TD error, loss, dL/dq  (idqn.py:111-124)
    float* delta = tA;  // [B][dm]
    for (int b = t; b < B; b += 256) {
        float tgt = a.reward[b] + (float)(1 - (int)a.terminal[b]) * a.gamma_n * qmax[b];
        int ac = a.action[b];
        float td = q[(long)b * dm + ac] - tgt;
        const float wgt = a.is_weight ? a.is_weight[b] : 1.0f;
        if (a.td_abs) a.td_abs[(long)k * B + b] = fabsf(td);
        sq[b] = wgt * td * td;
        for (int o = 0; o < A; ++o) delta[(long)b * dm + o] = (o == ac) ? 2.0f * wgt * td / (float)a.Bdiv : 0.f;
    }
    __syncthreads();
    if (t == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += sq[b];
        a.losses[k] = s / (float)a.Bdiv;
        const double tt = (double)(a.count[k] + 1);
        a.bcinv[2 * k] = 1.0f / (1.0f - (float)pow((double)a.adam_b1, tt));
        a.bcinv[2 * k + 1] = 1.0f / (1.0f - (float)pow((double)a.adam_b2, tt));
        if (a.finish_step) {
            a.count[k] += 1;
            a.cum[k] = a.cum[k] + (double)(s / (float)a.Bdiv);
        }
    }
    // ---- backward
    float* dprev = tB;
    for (int l = n.L - 1; l >= 0; --l) {
        const int din = n.d[l], dout = n.d[l + 1];
        const float* in = acts + (long)l * B * dm;
        const float* W = po + n.w_off[l];
        for (int e = t; e < din * dout; e += 256) {  // gW[i][o] = sum_b in[b][i] * delta[b][o]
            int i = e / dout, o = e - i * dout;
            float s = 0.f;
            for (int b = 0; b < B; ++b) s = fmaf(in[(long)b * dm + i], delta[(long)b * dm + o], s);
            G[n.w_off[l] + e] = s;
        }
        for (int o = t; o < dout; o += 256) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += delta[(long)b * dm + o];
            G[n.b_off[l] + o] = s;
        }
        if (l > 0) {
            for (int e = t; e < B * din; e += 256) {  // dprev[b][i] = relu'(in[b][i]) * sum_o delta[b][o] W[i][o]
                int b = e / din, i = e - b * din;
                float s = 0.f;
                const float* wr = W + (long)i * dout;
                const float* dr = delta + (long)b * dm;
                for (int o = 0; o < dout; ++o) s = fmaf(dr[o], wr[o], s);
                dprev[(long)b * dm + i] = in[(long)b * dm + i] > 0.f ? s : 0.f;
            }
        }
        __syncthreads();
        float* tmp = delta; delta = dprev; dprev = tmp;
    }
}

// Q-values of one net for n states (inference)
struct FcQArgs {
    FcNet net;
    const float* params;
    const float* s;  // [n][d0]
    float* ws;       // [2][n][dmax]
    float* q_out;    // [n][A]
    int n;
};
__global__ __launch_bounds__(256) void k_fc_q(FcQArgs a) {
    const FcNet& n = a.net;
    const int dm = n.dmax, t = threadIdx.x;
    float *cur = a.ws, *nxt = a.ws + (long)a.n * dm;
    for (int e = t; e < a.n * n.d[0]; e += 256) cur[(long)(e / n.d[0]) * dm + e % n.d[0]] = a.s[e];
    __syncthreads();
    for (int l = 0; l < n.L; ++l) {
        fc_layer(cur, dm, a.params + n.w_off[l], a.params + n.b_off[l], nxt, dm, a.n, n.d[l], n.d[l + 1], l != n.L - 1);
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    const int A = n.d[n.L];
    for (int e = t; e < a.n * A; e += 256) a.q_out[e] = cur[(long)(e / A) * dm + e % A];
}
