// gfx950 kernel of the MLP ("fc") i-DQN gradient step -- LunarLander-sized nets (8 -> 100 -> 100 -> 4).
//
// Reference: slimdqn/networks/architectures/dqn.py:61-70 (squeeze, [Dense + ReLU] x len(features), Dense)
// and slimdqn/networks/idqn.py:105-124.  These nets are ~11 k parameters per head: the step is
// launch-latency bound, so ONE workgroup per head runs the whole forward (online on s, target on s'),
// the TD target with the wavefront max over actions, the loss and the backward in a single launch;
// gradients go to the arena and the shared Adam kernel applies them.  Plain f32 FMAs: the layer
// widths (8, 100, 4) are not MFMA-shaped and the total is ~1 MFLOP per head.
#pragma once
#include <algorithm>

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
    // generic kernel (k_fc_step) only -- the dense head of the general-shape cnn path (gcnn_kernels.h):
    long s_stride;           // floats between the heads' own inputs in s / s2 (0: every head reads the same minibatch)
    float* din;              // [K][B][d0] out: dL/d(input), masked by input > 0 (the conv features are ReLU outputs), or nullptr
    GradMap gm;              // where a leaf's gradient lives in the arena
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
    const float* s_in = a.s + (long)k * a.s_stride;
    const float* s2_in = a.s2 + (long)k * a.s_stride;
    for (int e = t; e < B * n.d[0]; e += 256) tA[(long)(e / n.d[0]) * dm + e % n.d[0]] = s2_in[e];
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
    for (int e = t; e < B * n.d[0]; e += 256) acts[(long)(e / n.d[0]) * dm + e % n.d[0]] = s_in[e];
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
            a.grad[a.gm.at(k, n.w_off[l] + e)] = s;
        }
        for (int o = t; o < dout; o += 256) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += delta[(long)b * dm + o];
            a.grad[a.gm.at(k, n.b_off[l] + o)] = s;
        }
        if (l == 0 && a.din) {  // the conv trunk continues from here: dL/d(features), ReLU mask of the features included
            for (int e = t; e < B * din; e += 256) {
                int b = e / din, i = e - b * din;
                float s = 0.f;
                const float* wr = W + (long)i * dout;
                const float* dr = delta + (long)b * dm;
                for (int o = 0; o < dout; ++o) s = fmaf(dr[o], wr[o], s);
                a.din[((long)k * B + b) * din + i] = in[(long)b * dm + i] > 0.f ? s : 0.f;
            }
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

// ---------------------------------------------------------------------------------------------------
// LDS version of the step (used whenever the net fits: the LunarLander sizes need 118 KB).  The first version above
// kept every activation in a global workspace and walked each dot product as a chain of dependent L2 loads: 575 us per
// step for an 11 k-parameter head, slower than the Atari CNN.  Here a 512-thread workgroup per head keeps the
// activations of one 32-sample block, the deltas and the weight matrix of the current layer (odd row pitch) in LDS;
// every thread owns one output column and eight samples (or eight input rows in the weight gradient), so a weight is
// read once per eight FMAs and all LDS reads are broadcasts or unit-stride.  Batches larger than 32 are processed
// block by block; the gradient arena accumulates across blocks.
// ---------------------------------------------------------------------------------------------------
#define FC_T 512
#define FC_LDS_BUDGET (150 * 1024)  // bytes of LDS the step kernel may use

// Plan: samples per block BS (32, 16 or 8) and the width of the staged weight column tile, so that the activations,
// the deltas and one weight tile fit the budget.  Small nets get BS = 32 and whole matrices.
struct FcPlan {
    int BS, ld, tw, wfl;  // block size, activation pitch (odd), weight tile width, floats of the weight tile buffer
    long floats;          // total LDS floats
};
static inline FcPlan fc_plan(const FcNet& n) {
    FcPlan p;
    p.ld = n.dmax + 1 + (n.dmax & 1);
    int dinmax = 0, doutmax = 0;
    for (int l = 0; l < n.L; ++l) { dinmax = std::max(dinmax, n.d[l]); doutmax = std::max(doutmax, n.d[l + 1]); }
    for (int bs : {32, 16, 8}) {
        p.BS = bs;
        const long fixed = (long)(n.L + 3) * bs * p.ld + 3 * 32 + 8;  // acts + 2 delta buffers + qmax / sq / misc
        const long left = FC_LDS_BUDGET / 4 - fixed;
        if (left < (long)dinmax * 9) continue;  // not even an 8-column tile
        int tw = (int)std::min<long>(doutmax, left / dinmax - 1);
        tw = std::max(8, tw & ~1);              // even width -> odd pitch tw + 1
        p.tw = std::min(tw, doutmax + (doutmax & 1));
        p.wfl = dinmax * (p.tw | 1);
        p.floats = fixed + p.wfl;
        return p;
    }
    p.BS = 0; p.tw = 0; p.wfl = 0; p.floats = 0;  // does not fit: the generic kernel runs
    return p;
}

// stage columns [c0, c0 + cw) of W [din][dout] (global, row-major) into LDS with pitch (tw | 1)
__device__ __forceinline__ void fc_stage_w(const float* W, float* Wl, int din, int dout, int c0, int cw, int ldw) {
    for (int e = threadIdx.x; e < din * cw; e += FC_T) {
        const int i = e / cw, o = e - i * cw;
        Wl[i * ldw + o] = W[(long)i * dout + c0 + o];
    }
    __syncthreads();
}
// out[b][o] = (relu?)(bias[o] + sum_i in[b][i] * W[i][o]) for the BS samples of a block; in / out in LDS (pitch ld)
template <int BS>
__device__ __forceinline__ void fc_layer_lds(const float* in, const float* W, float* Wl, const float* bias, float* out,
                                             int ld, int din, int dout, bool relu, int tw) {
    constexpr int NG = BS / 8;
    const int ldw = tw | 1;
    for (int c0 = 0; c0 < dout; c0 += tw) {
        const int cw = min(tw, dout - c0);
        fc_stage_w(W, Wl, din, dout, c0, cw, ldw);
        for (int task = threadIdx.x; task < cw * NG; task += FC_T) {
            const int o = task % cw, bg = task / cw;
            const float* x = in + bg * 8 * ld;
            float acc[8];
            const float b0 = bias[c0 + o];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = b0;
            for (int i = 0; i < din; ++i) {
                const float w = Wl[i * ldw + o];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(x[j * ld + i], w, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) out[(bg * 8 + j) * ld + c0 + o] = relu ? fmaxf(acc[j], 0.f) : acc[j];
        }
        __syncthreads();
    }
}

template <int BS>
__global__ __launch_bounds__(FC_T) void k_fc_step_lds(FcArgs a, int tw, int wfl) {
    extern __shared__ __attribute__((aligned(16))) float fl[];
    constexpr int NG = BS / 8;
    const int k = blockIdx.x, t = threadIdx.x;
    const FcNet& n = a.net;
    const int B = a.B, A = n.d[n.L], ld = n.dmax + 1 + (n.dmax & 1);  // odd activation pitch
    float* Wl = fl;                               // staged weight tile of the current layer
    float* acts = Wl + wfl;                       // [L+1][BS][ld] online activations of the block
    float* dA = acts + (n.L + 1) * BS * ld;       // [BS][ld] target ping / delta
    float* dB = dA + BS * ld;                     // [BS][ld] target pong / delta
    float* qmax = dB + BS * ld;                   // [32]
    float* sq = qmax + 32;                        // [32]
    float* lsum = sq + 32;                        // [1]
    const float* po = a.online + (long)k * a.P;
    const float* pt = a.target + (long)k * a.P;
    float* G = a.grad + (long)k * a.P;
    if (t == 0) lsum[0] = 0.f;
    const int nb = (B + BS - 1) / BS;
    for (int bb = 0; bb < nb; ++bb) {
        const int b0 = bb * BS, nbk = min(BS, B - b0);
        // ---- inputs of the block (rows past the batch end are zero inputs; they carry no loss weight)
        for (int e = t; e < BS * n.d[0]; e += FC_T) {
            const int b = e / n.d[0], i = e - b * n.d[0];
            dA[b * ld + i] = b < nbk ? a.s2[(long)(b0 + b) * n.d[0] + i] : 0.f;
            acts[b * ld + i] = b < nbk ? a.s[(long)(b0 + b) * n.d[0] + i] : 0.f;
        }
        __syncthreads();
        // ---- target net on s'
        float *cur = dA, *nxt = dB;
        for (int l = 0; l < n.L; ++l) {
            fc_layer_lds<BS>(cur, pt + n.w_off[l], Wl, pt + n.b_off[l], nxt, ld, n.d[l], n.d[l + 1], l != n.L - 1, tw);
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
        if (t < 64) {  // wavefront max over actions: each half-wave folds every other action, one cross-lane step
            const int b = t & 31, hh = t >> 5;
            float m = -INFINITY;
            if (b < BS)
                for (int ac = hh; ac < A; ac += 2) m = fmaxf(m, cur[b * ld + ac]);
            m = fmaxf(m, __shfl_xor(m, 32));
            if (hh == 0 && b < BS) qmax[b] = m;
        }
        for (int e = t; e < nbk * A; e += FC_T) a.q_dbg[((long)(a.K + k) * B + b0) * A + e] = cur[(e / A) * ld + e % A];
        __syncthreads();
        // ---- online net on s, activations kept
        for (int l = 0; l < n.L; ++l)
            fc_layer_lds<BS>(acts + l * BS * ld, po + n.w_off[l], Wl, po + n.b_off[l], acts + (l + 1) * BS * ld, ld, n.d[l],
                             n.d[l + 1], l != n.L - 1, tw);
        const float* q = acts + n.L * BS * ld;
        for (int e = t; e < nbk * A; e += FC_T) a.q_dbg[((long)k * B + b0) * A + e] = q[(e / A) * ld + e % A];
        // ---- TD error, loss, dL/dq  (idqn.py:111-124)
        float* delta = dA;
        if (t < BS) {
            const int b = t;
            float sqv = 0.f, g = 0.f;
            int ac = 0;
            if (b < nbk) {
                const int bg = b0 + b;
                const float tgt = a.reward[bg] + (float)(1 - (int)a.terminal[bg]) * a.gamma_n * qmax[b];
                ac = a.action[bg];
                const float td = q[b * ld + ac] - tgt;
                const float wgt = a.is_weight ? a.is_weight[bg] : 1.0f;
                if (a.td_abs) a.td_abs[(long)k * B + bg] = fabsf(td);
                sqv = wgt * td * td;
                g = 2.0f * wgt * td / (float)a.Bdiv;
            }
            for (int o = 0; o < A; ++o) delta[b * ld + o] = (o == ac) ? g : 0.f;
            sq[b] = sqv;
        }
        __syncthreads();
        if (t == 0) {
            float s = lsum[0];
            for (int b = 0; b < BS; ++b) s += sq[b];
            lsum[0] = s;
        }
        // ---- backward of the block; the gradient arena accumulates over blocks
        float* dprev = dB;
        for (int l = n.L - 1; l >= 0; --l) {
            const int din = n.d[l], dout = n.d[l + 1], ldw = tw | 1;
            const float* in = acts + l * BS * ld;
            // gW[i][o] = sum_b in[b][i] * delta[b][o]: a thread owns column o and eight rows i
            const int n_ig = (din + 7) / 8;
            for (int task = t; task < dout * n_ig; task += FC_T) {
                const int o = task % dout, i0 = (task / dout) * 8;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int b = 0; b < BS; ++b) {
                    const float d = delta[b * ld + o];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(i0 + j < din ? in[b * ld + i0 + j] : 0.f, d, acc[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (i0 + j < din) {
                        float* g = G + n.w_off[l] + (long)(i0 + j) * dout + o;
                        *g = bb == 0 ? acc[j] : *g + acc[j];
                    }
            }
            for (int o = t; o < dout; o += FC_T) {
                float s = 0.f;
                for (int b = 0; b < BS; ++b) s += delta[b * ld + o];
                float* g = G + n.b_off[l] + o;
                *g = bb == 0 ? s : *g + s;
            }
            if (l > 0) {
                // dprev[b][i] = relu'(in[b][i]) * sum_o delta[b][o] W[i][o]: a thread owns row i and eight samples; the
                // sum runs over the weight column tiles, the partial sums stay in registers (<= 4 tasks per thread)
                float acc[4][8];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[q4][j] = 0.f;
                for (int c0 = 0; c0 < dout; c0 += tw) {
                    const int cw = min(tw, dout - c0);
                    fc_stage_w(po + n.w_off[l], Wl, din, dout, c0, cw, ldw);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int task = t + q4 * FC_T;
                        if (task < din * NG) {
                            const int i = task % din, bg = task / din;
                            const float* dr = delta + bg * 8 * ld + c0;
                            for (int o = 0; o < cw; ++o) {
                                const float w = Wl[i * ldw + o];
#pragma unroll
                                for (int j = 0; j < 8; ++j) acc[q4][j] = fmaf(dr[j * ld + o], w, acc[q4][j]);
                            }
                        }
                    }
                    __syncthreads();  // the tile buffer is re-staged next
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int task = t + q4 * FC_T;
                    if (task < din * NG) {
                        const int i = task % din, bg = task / din;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            dprev[(bg * 8 + j) * ld + i] = in[(bg * 8 + j) * ld + i] > 0.f ? acc[q4][j] : 0.f;
                    }
                }
            }
            __syncthreads();
            float* tmp = delta; delta = dprev; dprev = tmp;
        }
    }
    if (t == 0) {
        const float s = lsum[0];
        a.losses[k] = s / (float)a.Bdiv;
        const double tt = (double)(a.count[k] + 1);
        a.bcinv[2 * k] = 1.0f / (1.0f - (float)pow((double)a.adam_b1, tt));
        a.bcinv[2 * k + 1] = 1.0f / (1.0f - (float)pow((double)a.adam_b2, tt));
        if (a.finish_step) {
            a.count[k] += 1;
            a.cum[k] = a.cum[k] + (double)(s / (float)a.Bdiv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same step on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, k-ordered sums), for nets whose
// largest weight matrix fits LDS whole (the LunarLander sizes: 131 KB).  The LDS kernel above spends its time in LDS
// reads -- one weight and eight broadcast activations per eight FMAs: 92 us for an 11 k-parameter head.  Here every layer
// is a handful of 32 x 32 tiles:
//   activations and deltas are kept TRANSPOSED, [feature][32 samples] with a 33-float pitch, the staged weight matrix
//   [in][out] with an odd pitch: every MFMA operand is one conflict-free ds_read_b32 per lane, whichever of the three
//   contractions (forward: k = in; weight gradient: k = sample; data gradient: k = out) it serves;
//   tiles past a matrix edge read finite junk or staged zeros and their results are not written (the buffers are zeroed
//   once: nothing that is not finite ever gets into LDS, so 0 x junk stays 0).
// One 512-thread workgroup per head, tiles dealt round-robin to its 8 waves.
// ---------------------------------------------------------------------------------------------------
#define FCM_T 512
#define FCM_BSP 33  // sample pitch of the transposed activations
struct FcMfmaPlan {
    int ldw, drows;   // weight pitch (odd), rows of a delta / ping-pong buffer
    long w_floats, floats;  // floats of the weight buffer; total LDS floats (0: the net does not fit)
};
static inline FcMfmaPlan fc_mfma_plan(const FcNet& n) {
    FcMfmaPlan p;
    int dinmax = 0, doutmax = 0;
    for (int l = 0; l < n.L; ++l) { dinmax = std::max(dinmax, n.d[l]); doutmax = std::max(doutmax, n.d[l + 1]); }
    const int c32 = (doutmax + 31) / 32 * 32;
    p.ldw = c32 + 1;
    p.w_floats = (long)(dinmax + 1) * p.ldw;
    p.drows = (n.dmax + 31) / 32 * 32;
    // [weights][L + 1 activation buffers of dmax rows][2 delta buffers of drows rows][32 rows of slack for edge tiles][qmax, sq, misc]
    p.floats = p.w_floats + (long)(n.L + 1) * n.dmax * FCM_BSP + 2L * p.drows * FCM_BSP + 32L * FCM_BSP + 96;
    if (p.floats * 4 > FC_LDS_BUDGET) p.floats = 0;
    return p;
}

// The same kernel for nets whose weight matrix does NOT fit LDS beside the activations ([200, 200]: 181 KB): the weight
// operand of the forward / data-gradient tiles then comes straight from global memory (L2-resident: ~170 KB per head), in
// register chunks of FCM_CH MFMA steps requested one chunk ahead; activations and deltas stay in LDS, packed per layer
// (sum of the layer widths rows instead of (L + 1) x the widest).
#define FCM_CH 16
static inline FcMfmaPlan fc_mfma_plan_g(const FcNet& n) {
    FcMfmaPlan p;
    long rows = 0;
    for (int l = 0; l <= n.L; ++l) rows += n.d[l];
    p.ldw = 0;
    p.w_floats = 0;
    p.drows = (n.dmax + 31) / 32 * 32;
    p.floats = rows * FCM_BSP + 2L * p.drows * FCM_BSP + 32L * FCM_BSP + 96;
    if (p.floats * 4 > FC_LDS_BUDGET) p.floats = 0;
    return p;
}

// W [din][dout] (global) -> Wl [din + 1][ldw]: valid elements, zeros in the columns up to the next multiple of 32 and in
// row din (the k padding of an odd din / the n padding of the last column tile)
__device__ __forceinline__ void fcm_stage_w(const float* W, float* Wl, int din, int dout, int ldw) {
    const int c32 = (dout + 31) / 32 * 32;
    for (int e = threadIdx.x; e < (din + 1) * c32; e += FCM_T) {
        const int i = e / c32, o = e - i * c32;
        Wl[i * ldw + o] = (i < din && o < dout) ? W[(long)i * dout + o] : 0.f;
    }
    __syncthreads();
}

// outT[o][b] = (relu?)(bias[o] + sum_i inT[i][b] * W[i][o]): tiles of 32 columns o, k = i in pairs
__device__ __forceinline__ void fcm_forward(const float* inT, const float* Wl, const float* bias, float* outT, int din,
                                            int dout, int ldw, bool relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bl = lane & 31, h = lane >> 5;
    const int ks = (din + 1) / 2;
    for (int ct = wave; ct * 32 < dout; ct += FCM_T / 64) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* A = inT + h * FCM_BSP + bl;          // A[b = bl][k = 2 s + h]
        const float* Bp = Wl + h * ldw + ct * 32 + bl;    // B[k = 2 s + h][col = ct * 32 + bl]
        for (int s0 = 0; s0 < ks; ++s0) acc = mfma32(A[2 * s0 * FCM_BSP], Bp[2 * s0 * ldw], acc);
        const int col = ct * 32 + bl;
        if (col < dout) {
            const float bv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r] + bv;
                outT[col * FCM_BSP + mfma_row(r, h)] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
    __syncthreads();
}

// The forward with the weight operand read from global memory: lane (bl, h) of MFMA step s needs W[2 s + h][column]; a
// half-wave's 32 columns are one 128-byte line.  Columns past dout are clamped (their results are not written), rows past
// din read as zero.
__device__ __forceinline__ void fcm_forward_g(const float* inT, const float* W, const float* bias, float* outT, int din,
                                              int dout, bool relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bl = lane & 31, h = lane >> 5;
    const int ks = (din + 1) / 2, nch = (ks + FCM_CH - 1) / FCM_CH;
    for (int ct = wave; ct * 32 < dout; ct += FCM_T / 64) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* A = inT + h * FCM_BSP + bl;
        const float* Wc = W + min(ct * 32 + bl, dout - 1);
        float b0[FCM_CH], b1[FCM_CH];
#define FCM_LOAD(c, dst)                                                  \
    _Pragma("unroll") for (int u = 0; u < FCM_CH; ++u) {                  \
        const int kk = 2 * ((c) * FCM_CH + u) + h;                        \
        dst[u] = kk < din ? Wc[(long)kk * dout] : 0.f;                    \
    }
#define FCM_MMA(c, src)                                                   \
    _Pragma("unroll") for (int u = 0; u < FCM_CH; ++u) {                  \
        const int s0 = (c) * FCM_CH + u;                                  \
        if (s0 < ks) acc = mfma32(A[2 * s0 * FCM_BSP], src[u], acc);      \
    }
        FCM_LOAD(0, b0)
        for (int c = 0; c < nch; c += 2) {
            if (c + 1 < nch) { FCM_LOAD(c + 1, b1) }
            FCM_MMA(c, b0)
            if (c + 2 < nch) { FCM_LOAD(c + 2, b0) }
            if (c + 1 < nch) { FCM_MMA(c + 1, b1) }
        }
#undef FCM_LOAD
#undef FCM_MMA
        const int col = ct * 32 + bl;
        if (col < dout) {
            const float bv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r] + bv;
                outT[col * FCM_BSP + mfma_row(r, h)] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
    __syncthreads();
}

// WG = false: every weight matrix is staged whole in LDS (fc_mfma_plan); WG = true: weights from global (fc_mfma_plan_g)
template <bool WG>
__global__ __launch_bounds__(FCM_T) void k_fc_step_mfma(FcArgs a, int ldw, int drows, long w_floats) {
    extern __shared__ __attribute__((aligned(16))) float fl[];
    const int k = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, bl = lane & 31, h = lane >> 5;
    const FcNet& n = a.net;
    const int B = a.B, A = n.d[n.L], dm = n.dmax;
    float* Wl = fl;
    float* acts = Wl + w_floats;                         // [L + 1][dm][BSP]; WG: packed, layer l starts at row sum of d[0..l)
    auto act = [&](int l) {
        int row = l * dm;
        if (WG) {
            row = 0;
            for (int j = 0; j < l; ++j) row += n.d[j];
        }
        return acts + (long)row * FCM_BSP;
    };
    float* dA = act(n.L + 1);                            // [drows][BSP]
    float* dB = dA + (long)drows * FCM_BSP;
    float* qmax = dB + (long)(drows + 32) * FCM_BSP;     // behind the slack rows
    float* sq = qmax + 32;
    float* lsum = sq + 32;
    const float* po = a.online + (long)k * a.P;
    const float* pt = a.target + (long)k * a.P;
    float* G = a.grad + (long)k * a.P;
    for (long e = t; e < (qmax - fl) + 96; e += FCM_T) fl[e] = 0.f;  // nothing but finite numbers ever lives in this LDS
    __syncthreads();
    const int nb = (B + 31) / 32;
    for (int bb = 0; bb < nb; ++bb) {
        const int b0 = bb * 32, nbk = min(32, B - b0);
        // ---- inputs of the block, transposed (rows past the batch end are zero inputs; they carry no loss weight)
        for (int e = t; e < 32 * n.d[0]; e += FCM_T) {
            const int b = e / n.d[0], i = e - b * n.d[0];
            dA[i * FCM_BSP + b] = b < nbk ? a.s2[(long)(b0 + b) * n.d[0] + i] : 0.f;
            acts[i * FCM_BSP + b] = b < nbk ? a.s[(long)(b0 + b) * n.d[0] + i] : 0.f;
        }
        __syncthreads();
        // ---- target net on s'
        float *cur = dA, *nxt = dB;
        for (int l = 0; l < n.L; ++l) {
            if (WG) {
                fcm_forward_g(cur, pt + n.w_off[l], pt + n.b_off[l], nxt, n.d[l], n.d[l + 1], l != n.L - 1);
            } else {
                fcm_stage_w(pt + n.w_off[l], Wl, n.d[l], n.d[l + 1], ldw);
                fcm_forward(cur, Wl, pt + n.b_off[l], nxt, n.d[l], n.d[l + 1], ldw, l != n.L - 1);
            }
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
        if (t < 32) {  // max over actions, in action order
            float m = -INFINITY;
            for (int ac = 0; ac < A; ++ac) m = fmaxf(m, cur[ac * FCM_BSP + t]);
            qmax[t] = m;
        }
        for (int e = t; e < nbk * A; e += FCM_T) a.q_dbg[((long)(a.K + k) * B + b0) * A + e] = cur[(e % A) * FCM_BSP + e / A];
        __syncthreads();
        // ---- online net on s, activations kept
        for (int l = 0; l < n.L; ++l) {
            if (WG) {
                fcm_forward_g(act(l), po + n.w_off[l], po + n.b_off[l], act(l + 1), n.d[l], n.d[l + 1], l != n.L - 1);
            } else {
                fcm_stage_w(po + n.w_off[l], Wl, n.d[l], n.d[l + 1], ldw);
                fcm_forward(act(l), Wl, po + n.b_off[l], act(l + 1), n.d[l], n.d[l + 1], ldw, l != n.L - 1);
            }
        }
        const float* q = act(n.L);  // [A][BSP]
        for (int e = t; e < nbk * A; e += FCM_T) a.q_dbg[((long)k * B + b0) * A + e] = q[(e % A) * FCM_BSP + e / A];
        // ---- TD error, loss, dL/dq  (idqn.py:111-124)
        float* delta = dA;
        if (t < 32) {
            const int b = t;
            float sqv = 0.f, g = 0.f;
            int ac = 0;
            if (b < nbk) {
                const int bg = b0 + b;
                const float tgt = a.reward[bg] + (float)(1 - (int)a.terminal[bg]) * a.gamma_n * qmax[b];
                ac = a.action[bg];
                const float td = q[ac * FCM_BSP + b] - tgt;
                const float wgt = a.is_weight ? a.is_weight[bg] : 1.0f;
                if (a.td_abs) a.td_abs[(long)k * B + bg] = fabsf(td);
                sqv = wgt * td * td;
                g = 2.0f * wgt * td / (float)a.Bdiv;
            }
            for (int o = 0; o < A; ++o) delta[o * FCM_BSP + b] = (o == ac) ? g : 0.f;
            sq[b] = sqv;
        }
        __syncthreads();
        if (t == 0) {
            float s = lsum[0];
            for (int b = 0; b < 32; ++b) s += sq[b];
            lsum[0] = s;
        }
        // ---- backward of the block; the gradient arena accumulates over blocks.  Wl holds the online matrix of layer L - 1.
        float* dprev = dB;
        for (int l = n.L - 1; l >= 0; --l) {
            const int din = n.d[l], dout = n.d[l + 1];
            const float* inT = act(l);
            if (!WG && l != n.L - 1) fcm_stage_w(po + n.w_off[l], Wl, din, dout, ldw);
            // gW[i][o] = sum_b inT[i][b] * delta[o][b]: tiles of 32 rows i x 32 columns o, k = the 32 samples
            const int nti = (din + 31) / 32, nto = (dout + 31) / 32;
            for (int tile = wave; tile < nti * nto; tile += FCM_T / 64) {
                const int ti = tile / nto, to = tile - ti * nto;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* Ap = inT + (long)(ti * 32 + bl) * FCM_BSP + h;    // A[i = bl][k = b = 2 s + h]
                const float* Bp = delta + (long)(to * 32 + bl) * FCM_BSP + h;  // B[k = b][o = bl]
#pragma unroll
                for (int s0 = 0; s0 < 16; ++s0) acc = mfma32(Ap[2 * s0], Bp[2 * s0], acc);
                const int o = to * 32 + bl;
                if (o < dout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = ti * 32 + mfma_row(r, h);
                        if (i < din) {
                            float* g = G + n.w_off[l] + (long)i * dout + o;
                            *g = bb == 0 ? acc[r] : *g + acc[r];
                        }
                    }
                }
            }
            for (int o = t; o < dout; o += FCM_T) {  // bias gradient, in sample order
                float s = 0.f;
                for (int b = 0; b < 32; ++b) s += delta[o * FCM_BSP + b];
                float* g = G + n.b_off[l] + o;
                *g = bb == 0 ? s : *g + s;
            }
            if (l > 0) {
                // dprev[i][b] = relu'(in[i][b]) * sum_o delta[o][b] * W[i][o]: tiles of 32 columns i, k = o in pairs
                const int ks = (dout + 1) / 2;
                for (int ti = wave; ti * 32 < din; ti += FCM_T / 64) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    const float* Ap = delta + h * FCM_BSP + bl;                  // A[b = bl][k = o = 2 s + h]
                    if (WG) {
                        // B[k = o][i = bl] = W[i][o] from global: a lane walks its own row (rows past din clamped, not
                        // written), 2 s + h; a 128-byte line serves 16 steps of both half-waves from L1
                        const float* Wr = po + n.w_off[l] + (long)min(ti * 32 + bl, din - 1) * dout;
                        const int nch = (ks + FCM_CH - 1) / FCM_CH;
                        float b0[FCM_CH], b1[FCM_CH];
#define FCM_LOAD(c, dst)                                                  \
    _Pragma("unroll") for (int u = 0; u < FCM_CH; ++u) {                  \
        const int o = 2 * ((c) * FCM_CH + u) + h;                         \
        dst[u] = o < dout ? Wr[o] : 0.f;                                  \
    }
#define FCM_MMA(c, src)                                                   \
    _Pragma("unroll") for (int u = 0; u < FCM_CH; ++u) {                  \
        const int s0 = (c) * FCM_CH + u;                                  \
        if (s0 < ks) acc = mfma32(Ap[2 * s0 * FCM_BSP], src[u], acc);     \
    }
                        FCM_LOAD(0, b0)
                        for (int c = 0; c < nch; c += 2) {
                            if (c + 1 < nch) { FCM_LOAD(c + 1, b1) }
                            FCM_MMA(c, b0)
                            if (c + 2 < nch) { FCM_LOAD(c + 2, b0) }
                            if (c + 1 < nch) { FCM_MMA(c + 1, b1) }
                        }
#undef FCM_LOAD
#undef FCM_MMA
                    } else {
                        const float* Bp = Wl + (long)(ti * 32 + bl) * ldw + h;   // B[k = o][i = bl]   (rows past din: junk, not written)
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
                }
            }
            __syncthreads();
            float* tmp = delta; delta = dprev; dprev = tmp;
        }
    }
    if (t == 0) {
        const float s = lsum[0];
        a.losses[k] = s / (float)a.Bdiv;
        const double tt = (double)(a.count[k] + 1);
        a.bcinv[2 * k] = 1.0f / (1.0f - (float)pow((double)a.adam_b1, tt));
        a.bcinv[2 * k + 1] = 1.0f / (1.0f - (float)pow((double)a.adam_b2, tt));
        if (a.finish_step) {
            a.count[k] += 1;
            a.cum[k] = a.cum[k] + (double)(s / (float)a.Bdiv);
        }
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

// One state (acting): the same forward with the k index of every layer cut into 8 slices, one per wave -- a lane's loads of
// a slice (<= 64 rows) are all in flight at once and coalesced along the output row; the slices are added in order.
// (k_fc_q walks each dot product as one chain of dependent fmas over global loads: 19 us for the LunarLander net.)
__global__ __launch_bounds__(512) void k_fc_q1(FcQArgs a) {
    __shared__ float x[2][FC_MAX_WIDTH];
    __shared__ float part[8][FC_MAX_WIDTH];
    const FcNet& n = a.net;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int e = t; e < n.d[0]; e += 512) x[0][e] = a.s[e];
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < n.L; ++l) {
        const int din = n.d[l], dout = n.d[l + 1];
        const float* W = a.params + n.w_off[l];
        const int per = (din + 7) / 8, i0 = wave * per, i1 = min(din, i0 + per);  // per <= 64 (widths <= 512)
        for (int o = lane; o < dout; o += 64) {
            float s = 0.f;
            for (int ib = i0; ib < i1; ib += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = ib + u < i1 ? W[(long)(ib + u) * dout + o] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) s = fmaf(ib + u < i1 ? x[cur][ib + u] : 0.f, w[u], s);
            }
            part[wave][o] = s;
        }
        __syncthreads();
        for (int o = t; o < dout; o += 512) {
            float s = a.params[n.b_off[l] + o];
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) s += part[w8][o];
            x[cur ^ 1][o] = l != n.L - 1 ? fmaxf(s, 0.f) : s;
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int e = t; e < n.d[n.L]; e += 512) a.q_out[e] = x[cur][e];
}
