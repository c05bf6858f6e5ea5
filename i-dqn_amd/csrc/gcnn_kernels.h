// General-shape cnn path: plain f32 HIP kernels for every `features` / channel / action count the reference's DQNNet
// accepts (slimdqn/networks/architectures/dqn.py:39-53,65-70 -- e.g. the reference's own smoke test trains
// `--features 2 3 1 15`, tests/test_atari.py:24-28), used whenever a shape is outside what the MFMA plane kernels are
// built for (conv widths 32 / 64, 4 input channels, one hidden dense layer of 128..512, <= 32 actions).
// Correct first, not fast: one thread per output element, sequential (deterministic) sums, NHWC f32 activations
// [net][sample][h][w][c]; the dense head runs on the generic MLP step kernel (fc_kernels.h, k_fc_step) with per-head
// inputs = the flattened conv features.  Still the HIP path: there is no host fallback for any shape.
#pragma once
#include "common.h"

struct GConvArgs {
    const uint8_t* in_u8[2];    // layer 0: the two uint8 minibatches (state, next_state) [B][IH][IW][CI]; nets < n_split read [0]
    const float* in;            // layers 1, 2: [n_nets][B][IH][IW][CI]
    float* out;                 // [n_nets][B][OH][OW][CO]  relu(conv + bias)
    const float* const* wbase;  // [n_nets]
    long w_off, b_off;
    int n_nets, n_split, B, IH, IW, CI, OH, OW, CO, KS, S, PLh, PLw;
};

__device__ __forceinline__ float gconv_in(const GConvArgs& a, int net, int b, int ih, int iw, int ci) {
    if (a.in) return a.in[((((long)net * a.B + b) * a.IH + ih) * a.IW + iw) * a.CI + ci];
    const uint8_t* s = a.in_u8[net < a.n_split ? 0 : 1];
    return (float)s[(((long)b * a.IH + ih) * a.IW + iw) * a.CI + ci] / 255.0f;  // architectures/dqn.py:44
}

__global__ __launch_bounds__(256) void k_gconv_fwd(GConvArgs a) {
    const long n = (long)a.n_nets * a.B * a.OH * a.OW * a.CO;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        long r = e;
        const int co = (int)(r % a.CO); r /= a.CO;
        const int ow = (int)(r % a.OW); r /= a.OW;
        const int oh = (int)(r % a.OH); r /= a.OH;
        const int b = (int)(r % a.B);
        const int net = (int)(r / a.B);
        const float* P = a.wbase[net];
        const float* W = P + a.w_off;
        float s = P[a.b_off + co];
        for (int kh = 0; kh < a.KS; ++kh) {
            const int ih = oh * a.S + kh - a.PLh;
            if (ih < 0 || ih >= a.IH) continue;
            for (int kw = 0; kw < a.KS; ++kw) {
                const int iw = ow * a.S + kw - a.PLw;
                if (iw < 0 || iw >= a.IW) continue;
                for (int ci = 0; ci < a.CI; ++ci)
                    s = fmaf(gconv_in(a, net, b, ih, iw, ci), W[((long)(kh * a.KS + kw) * a.CI + ci) * a.CO + co], s);
            }
        }
        a.out[e] = fmaxf(s, 0.f);
    }
}

struct GConvBwdArgs {
    GConvArgs f;        // the forward geometry of the layer (n_nets = K online nets; in / in_u8 = its forward input)
    const float* dy;    // [K][B][OH][OW][CO]  gradient w.r.t. the layer's PRE-activation (ReLU mask already applied)
    float* din;         // dgrad: [K][B][IH][IW][CI], masked by the forward input > 0 (= ReLU of the layer below)
    float* grad;        // wgrad: the gradient arena
    GradMap gm;
};

// weight + bias gradient: one thread per element of the HWIO kernel (then per bias element)
__global__ __launch_bounds__(256) void k_gconv_wgrad(GConvBwdArgs g) {
    const GConvArgs& a = g.f;
    const long nw = (long)a.KS * a.KS * a.CI * a.CO;
    const int k = blockIdx.y;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nw + a.CO; e += (long)gridDim.x * 256) {
        float s = 0.f;
        if (e < nw) {
            long r = e;
            const int co = (int)(r % a.CO); r /= a.CO;
            const int ci = (int)(r % a.CI); r /= a.CI;
            const int kw = (int)(r % a.KS);
            const int kh = (int)(r / a.KS);
            for (int b = 0; b < a.B; ++b)
                for (int oh = 0; oh < a.OH; ++oh) {
                    const int ih = oh * a.S + kh - a.PLh;
                    if (ih < 0 || ih >= a.IH) continue;
                    for (int ow = 0; ow < a.OW; ++ow) {
                        const int iw = ow * a.S + kw - a.PLw;
                        if (iw < 0 || iw >= a.IW) continue;
                        s = fmaf(gconv_in(a, k, b, ih, iw, ci), g.dy[((((long)k * a.B + b) * a.OH + oh) * a.OW + ow) * a.CO + co], s);
                    }
                }
            g.grad[g.gm.at(k, a.w_off + e)] = s;
        } else {
            const int co = (int)(e - nw);
            for (long p = 0; p < (long)a.B * a.OH * a.OW; ++p) s += g.dy[((long)k * a.B * a.OH * a.OW + p) * a.CO + co];
            g.grad[g.gm.at(k, a.b_off + co)] = s;
        }
    }
}

// data gradient w.r.t. the layer's input, times the ReLU mask of that input (the activation of the layer below)
__global__ __launch_bounds__(256) void k_gconv_dgrad(GConvBwdArgs g) {
    const GConvArgs& a = g.f;
    const long n = (long)a.n_nets * a.B * a.IH * a.IW * a.CI;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        long r = e;
        const int ci = (int)(r % a.CI); r /= a.CI;
        const int iw = (int)(r % a.IW); r /= a.IW;
        const int ih = (int)(r % a.IH); r /= a.IH;
        const int b = (int)(r % a.B);
        const int k = (int)(r / a.B);
        float s = 0.f;
        if (a.in[e] > 0.f) {
            const float* W = a.wbase[k] + a.w_off;
            for (int kh = 0; kh < a.KS; ++kh) {
                const int th = ih + a.PLh - kh;
                if (th < 0 || th % a.S) continue;
                const int oh = th / a.S;
                if (oh >= a.OH) continue;
                for (int kw = 0; kw < a.KS; ++kw) {
                    const int tw = iw + a.PLw - kw;
                    if (tw < 0 || tw % a.S) continue;
                    const int ow = tw / a.S;
                    if (ow >= a.OW) continue;
                    const float* dy = g.dy + ((((long)k * a.B + b) * a.OH + oh) * a.OW + ow) * a.CO;
                    const float* w = W + ((long)(kh * a.KS + kw) * a.CI + ci) * a.CO;
                    for (int co = 0; co < a.CO; ++co) s = fmaf(dy[co], w[co], s);
                }
            }
        }
        g.din[e] = s;
    }
}
