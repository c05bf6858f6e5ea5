// Acting path for ONE state (select_action / best_action: slimdqn/sample_collection/utils.py:8-21, idqn.py:126-131).
//
// The training kernels are built around 32-sample MFMA tiles and one-workgroup-per-CU launches; for a single state that
// is 31/32 wasted work behind ~60 us of kernel time (7 launches, kernel packing included).  A forward pass of one state
// is 16 MMAC -- nothing -- so this path is about latency only: five small f32-FMA launches that read the uint8 pixels and
// the HWIO / [in][out] parameter leaves as they are (no staging, no packing), SAME padding by bounds checks, the k index
// split over the lanes of a workgroup so that no thread walks more than ~72 dependent FMAs, and the 15.9 MB Dense_0 kernel
// streamed once by F / 32 workgroups.  x / 255 as in architectures/dqn.py:44; sums are plain f32 fma chains.
#pragma once
#include "common.h"

struct ActConvArgs {
    const uint8_t* in_u8;  // layer 0: [IH][IW][CI] uint8
    const float* in;       // later layers: [IH][IW][CI] f32
    const float* params;   // the net's parameter base
    float* out;            // [OH][OW][CO] f32, after bias + ReLU
    long w_off, b_off;
    int IH, IW, CI, OH, OW, CO, K, S, PLh, PLw;
    int KS;                // k-splits: lanes per output; 256 / KS outputs per workgroup
};

// workgroup = 256 / KS consecutive outputs (position-major, channel-minor) x KS lane slices.  The (kh, kw, ci) range is
// cut into units of one tap x CIU channels (contiguous in the input, stride CO in the HWIO leaf); slice ks takes units
// ks, ks + KS, ...; a unit's CIU loads are all in flight before its fma chain runs.  Layer 0 reads uint8 through a
// 256-entry table of u / 255 (correctly rounded, as the reference's division) built once per workgroup.
template <int CIU, int UPT>  // channels per unit; units per thread (>= ceil(K * K * CI / CIU / KS))
__global__ __launch_bounds__(256) void k_act_conv(ActConvArgs a) {
    __shared__ float red[256];
    __shared__ float lut[256];
    const int t = threadIdx.x, per = 256 / a.KS, oi = t % per, ks = t / per;
    if (a.in_u8) {
        lut[t] = (float)t / 255.0f;
        __syncthreads();
    }
    const long o = (long)blockIdx.x * per + oi;
    const int n_out = a.OH * a.OW * a.CO, ccs = a.CI / CIU, n_units = a.K * a.K * ccs;
    float acc = 0.f;
    if (o < n_out) {
        const int co = (int)(o % a.CO), pos = (int)(o / a.CO), oh = pos / a.OW, ow = pos - oh * a.OW;
        const float* W = a.params + a.w_off + co;
        float wv[UPT][CIU], xv[UPT][CIU];
#pragma unroll
        for (int n = 0; n < UPT; ++n) {  // every load of every unit is issued before the first fma
            const int u = ks + n * a.KS, uc = min(u, n_units - 1);
            const int tap = uc / ccs, c0 = (uc - tap * ccs) * CIU, kh = tap / a.K, kw = tap - kh * a.K;
            const int ih = oh * a.S + kh - a.PLh, iw = ow * a.S + kw - a.PLw;
            const bool live = u < n_units && ih >= 0 && ih < a.IH && iw >= 0 && iw < a.IW;  // SAME padding: zeros
            const long xi = live ? ((long)ih * a.IW + iw) * a.CI + c0 : 0;
            const float* w = W + ((long)tap * a.CI + c0) * a.CO;
#pragma unroll
            for (int ci = 0; ci < CIU; ++ci) wv[n][ci] = live ? w[(long)ci * a.CO] : 0.f;
            if (a.in_u8) {
                if (CIU == 4) {  // 4 channels = one aligned 32-bit load
                    const unsigned px = *reinterpret_cast<const unsigned*>(a.in_u8 + xi);
#pragma unroll
                    for (int ci = 0; ci < CIU; ++ci) xv[n][ci] = lut[(px >> (8 * ci)) & 0xffu];
                } else {
#pragma unroll
                    for (int ci = 0; ci < CIU; ++ci) xv[n][ci] = lut[a.in_u8[xi + ci]];
                }
            } else {
#pragma unroll
                for (int ci = 0; ci < CIU; ++ci) xv[n][ci] = a.in[xi + ci];
            }
        }
#pragma unroll
        for (int n = 0; n < UPT; ++n)
#pragma unroll
            for (int ci = 0; ci < CIU; ++ci) acc = fmaf(xv[n][ci], wv[n][ci], acc);
    }
    red[t] = acc;
    __syncthreads();
    if (ks == 0 && o < n_out) {
        float s = red[oi];
        for (int j = 1; j < a.KS; ++j) s += red[j * per + oi];  // slices in order
        const int co = (int)(o % a.CO);
        a.out[o] = fmaxf(s + a.params[a.b_off + co], 0.f);
    }
}

struct ActDenseArgs {
    const float* a3;      // [F]
    const float* params;
    float* part;          // [NRG][J]
    long w_off;
    int F, J, NRG;        // NRG row groups
};
// workgroup = (row group rg of F / NRG rows, 128-column quarter): 8 row sub-groups x 32 float4 column quads; a thread's
// rows are all in flight at once, the 8 sub-groups are added in order through LDS.  NRG x J / 128 workgroups stream the
// 15.9 MB kernel; only NRG partial rows are left for the head (one CU reads ~10 B/clk: 242 rows cost it 12 us).
__global__ __launch_bounds__(256) void k_act_dense0(ActDenseArgs a) {
    __shared__ float4 red[8][32];
    const int t = threadIdx.x, cq = t & 31, rs = t >> 5, nq = a.J / 128;
    const int rg = blockIdx.x / nq, jq = (blockIdx.x - rg * nq) * 128 + cq * 4;
    const int r0 = (int)((long)a.F * rg / a.NRG), r1 = (int)((long)a.F * (rg + 1) / a.NRG);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rb = r0 + rs; rb < r1; rb += 8 * 16) {  // 16 rows per thread and round
        float4 w[16];
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = min(rb + 8 * u, r1 - 1);
            w[u] = *reinterpret_cast<const float4*>(a.params + a.w_off + (long)r * a.J + jq);
            x[u] = a.a3[r];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (rb + 8 * u < r1) {
                s.x = fmaf(x[u], w[u].x, s.x); s.y = fmaf(x[u], w[u].y, s.y);
                s.z = fmaf(x[u], w[u].z, s.z); s.w = fmaf(x[u], w[u].w, s.w);
            }
    }
    red[rs][cq] = s;
    __syncthreads();
    if (rs == 0) {
        float4 v = red[0][cq];
#pragma unroll
        for (int j = 1; j < 8; ++j) { const float4 y = red[j][cq]; v.x += y.x; v.y += y.y; v.z += y.z; v.w += y.w; }
        *reinterpret_cast<float4*>(a.part + (long)rg * a.J + jq) = v;
    }
}

struct ActHeadArgs {
    const float* part;  // [NP][J]
    const float* params;
    long b0_off, w1_off, b1_off;
    int NP, J, A;
    float* q_out;       // [A]
    int32_t* action;    // [1] or nullptr
    // host mailbox (idqn_act_host) or nullptr: {action, sequence number} in mapped, coherent host memory.  The kernel
    // bumps the device-side counter and writes the action, then the number; the host polls the number instead of paying
    // for a device-to-host copy node and a stream synchronisation.
    volatile int32_t* mail;
    unsigned* seq;
};
// one workgroup of 1024: h = relu(b0 + sum of the partials), q = b1 + h W1, first maximum (jnp.argmax).  The NP = F / 32
// partial rows are summed by 4 row groups x 256 column pairs (float2, 32 loads in flight), groups combined in order.
__global__ __launch_bounds__(1024) void k_act_head(ActHeadArgs a) {
    __shared__ float hp[4][512];
    __shared__ float hs[512];
    __shared__ float qs[32];
    const int t = threadIdx.x, g = t >> 8, jp = (t & 255) * 2;
    if (jp < a.J) {
        const int p0 = a.NP * g / 4, p1 = a.NP * (g + 1) / 4;
        float sx = 0.f, sy = 0.f;
        for (int p = p0; p < p1; p += 32) {  // (clamped: the tail re-reads the group's last row)
            float2 v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = *reinterpret_cast<const float2*>(a.part + (long)min(p + u, p1 - 1) * a.J + jp);
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (p + u < p1) { sx += v[u].x; sy += v[u].y; }
        }
        hp[g][jp] = sx;
        hp[g][jp + 1] = sy;
    }
    __syncthreads();
    if (t < a.J) hs[t] = fmaxf(((hp[0][t] + hp[1][t]) + (hp[2][t] + hp[3][t])) + a.params[a.b0_off + t], 0.f);
    __syncthreads();
    const int wave = t >> 6, lane = t & 63;
    for (int ac = wave; ac < a.A; ac += 16) {  // one wave per action: lanes stride the hidden units, then a wave reduction
        float s = 0.f, wv[8];  // J <= 512: at most 8 hidden units per lane, their weights all requested before the chain
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = lane + 64 * u < a.J ? a.params[a.w1_off + (long)(lane + 64 * u) * a.A + ac] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(lane + 64 * u < a.J ? hs[lane + 64 * u] : 0.f, wv[u], s);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const float q = s + a.params[a.b1_off + ac];
            qs[ac] = q;
            a.q_out[ac] = q;
        }
    }
    __syncthreads();
    if (t == 0 && a.action) {
        int best = 0;
        float bv = qs[0];
        for (int ac = 1; ac < a.A; ++ac)
            if (qs[ac] > bv) { bv = qs[ac]; best = ac; }
        a.action[0] = best;
        if (a.mail) {
            const unsigned n = a.seq[0] + 1u;
            a.seq[0] = n;
            a.mail[0] = best;
            __threadfence_system();  // the action is visible to the host before the number that announces it
            a.mail[1] = (int32_t)n;
        }
    }
}
