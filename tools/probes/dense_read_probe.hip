// How fast can the Dense_0 forward's ACCESS PATTERN stream its 158.6 MB of weights, without its arithmetic?  Emulates
// k_dense0_fwd3's loads (10 nets x [7744][512] f32; a wave = (net, split, 128-column tile), lane (bl, h) reads float4 of
// rows 16 c + 8 h + jj; interleaved split-K over NS splits; ring of AHEAD + 1 k-steps) next to a plain grid-stride read.
// hipcc --offload-arch=gfx950 -O3 dense_read_probe.hip -o dense_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int F = 7744, J = 512, NETS = 10;

template <int AHEAD, bool WITH_X, int MODE>  // MODE 0: wave = 128-col tile (float4 per lane, 2 rows per instruction);
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ Wb, const float* __restrict__ Xb, int NS, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bl = lane & 31, h = lane >> 5;
    long item = (long)blockIdx.x * 4 + wave;
    const int jt = (int)(item % 4);
    item /= 4;
    const int s = (int)(item % NS), n = (int)(item / NS);
    if (n >= NETS) return;
    const int NU = F / 16, NC = (NU - s + NS - 1) / NS;
    const long step_rows = 16L * NS;
    const float* W;
    if (MODE == 0) W = Wb + (long)n * F * J + (long)(16 * s + 8 * h) * J + jt * 128 + 4 * bl;
    else W = Wb + (long)n * F * J + (long)(16 * s + 4 * jt) * J + lane * 4;  // MODE 1: a wave reads 4 whole rows (2 KB each): lane = 16-byte column
    const float* X = Xb + (long)n * F * 32 + (long)(16 * s + 8 * h) * 32 + bl;
    float acc = 0.f;
    constexpr int R = AHEAD + 1;
    float4 wv[R][8];
    float xv[R][8];
    auto load = [&](int c, int slot) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            if (MODE == 0) wv[slot][jj] = *reinterpret_cast<const float4*>(W + ((long)c * step_rows + jj) * J);
            else wv[slot][jj] = *reinterpret_cast<const float4*>(W + (long)c * step_rows * J + (jj >> 1) * J + (jj & 1) * 256);
            if (WITH_X) xv[slot][jj] = X[((long)c * step_rows + jj) * 32];
        }
    };
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) load(u < NC ? u : NC - 1, u);
    for (int c = 0; c < NC; c += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int cn = c + u + AHEAD;
            load(cn < NC ? cn : NC - 1, (u + AHEAD) % R);
            __builtin_amdgcn_sched_barrier(0);
            if (c + u < NC) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) acc += wv[u][jj].x + wv[u][jj].y + wv[u][jj].z + wv[u][jj].w + (WITH_X ? xv[u][jj] : 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, long n4, float* out) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void k_thrash(float4* p, long n4) {  // evicts the caches between timed launches (the step touches 570 MB)
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    const long wbytes = (long)NETS * F * J * 4, xbytes = (long)NETS * F * 32 * 4, tbytes = 600L << 20;
    float *W, *X, *out; float4* T;
    CK(hipMalloc(&W, wbytes)); CK(hipMalloc(&X, xbytes)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&T, tbytes));
    CK(hipMemset(W, 1, wbytes)); CK(hipMemset(X, 1, xbytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 6; ++rep) {
            hipLaunchKernelGGL(k_thrash, dim3(1024), dim3(256), 0, 0, T, tbytes / 16);
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; if (rep) sum += ms;
        }
        printf("%-44s best %6.1f us  mean %6.1f us  %.2f TB/s (W bytes / mean)\n", name, best * 1e3, sum / 5 * 1e3, wbytes / (sum / 5) / 1e9);
    };
    timeit("plain grid-stride read, 256 WGs", [&] { hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, (const float4*)W, wbytes / 16, out); });
    timeit("plain grid-stride read, 512 WGs", [&] { hipLaunchKernelGGL(k_read, dim3(512), dim3(256), 0, 0, (const float4*)W, wbytes / 16, out); });
    for (int NS : {25, 50, 100}) {
        char nm[96];
        const int wgs = NETS * NS;
        snprintf(nm, 96, "fwd pattern NS=%d ahead 3 with X", NS);  timeit(nm, [&] { hipLaunchKernelGGL((k_rows<3, true, 0>), dim3(wgs), dim3(256), 0, 0, W, X, NS, out); });
        snprintf(nm, 96, "fwd pattern NS=%d ahead 3 no X", NS);    timeit(nm, [&] { hipLaunchKernelGGL((k_rows<3, false, 0>), dim3(wgs), dim3(256), 0, 0, W, X, NS, out); });
        snprintf(nm, 96, "fwd pattern NS=%d ahead 1 no X", NS);    timeit(nm, [&] { hipLaunchKernelGGL((k_rows<1, false, 0>), dim3(wgs), dim3(256), 0, 0, W, X, NS, out); });
        snprintf(nm, 96, "whole-row waves NS=%d ahead 3 no X", NS); timeit(nm, [&] { hipLaunchKernelGGL((k_rows<3, false, 1>), dim3(wgs), dim3(256), 0, 0, W, X, NS, out); });
        snprintf(nm, 96, "whole-row waves NS=%d ahead 1 no X", NS); timeit(nm, [&] { hipLaunchKernelGGL((k_rows<1, false, 1>), dim3(wgs), dim3(256), 0, 0, W, X, NS, out); });
    }
    return 0;
}
