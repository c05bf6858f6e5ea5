// Can the Dense_0 forward stream its weights by COLUMN tiles instead of by k-splits?  A workgroup = (net, 32-column tile) needs no
// cross-workgroup reduction (no partial slabs, no k_hidden launch), but reads 128-byte pieces of 2 KB rows: 160 workgroups of NW
// waves, wave w takes the rows [w R / NW, (w + 1) R / NW); one wave-instruction = 8 rows x 128 B.  Against the k-split pattern of
// k_dense0_fwd3 (250 workgroups x 4 waves, 512-byte pieces, interleaved 16-row steps).  Optional second stream: the activation
// fragments of the k-step from an L2-resident buffer (3 x 1 KB per 16 rows), as the real kernel would read them.
// hipcc --offload-arch=gfx950 -O3 d0_coltile_probe.hip -o d0_coltile_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define NETS 10
#define F 7744
#define J 512

template <int NW, bool XS>
__global__ __launch_bounds__(NW * 64) void k_col(const float* __restrict__ w, const float* __restrict__ x, float* out) {
    const int net = blockIdx.x / (J / 32), jt = blockIdx.x % (J / 32);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = (int)((long)F * wave / NW), r1 = (int)((long)F * (wave + 1) / NW);
    const float* p = w + ((long)net * F + r0 + (lane >> 3)) * J + jt * 32 + (lane & 7) * 4;
    const float* xp = x + (long)net * F * 24 + (long)r0 * 24 + lane * 4;  // 3 KB per 16 rows = 48 floats per row pair... (fragment-ordered)
    float s = 0.f;
    int r = r0;
    for (; r + 64 <= r1; r += 64) {  // 8 wave-loads of 8 rows in flight
        f32x4v v[8], xv[6];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p + (long)(r - r0 + 8 * u) * J));
        if (XS) {
#pragma unroll
            for (int u = 0; u < 6; ++u) xv[u] = *reinterpret_cast<const f32x4v*>(xp + (long)(r - r0) * 24 + u * 256);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
        if (XS) {
#pragma unroll
            for (int u = 0; u < 6; ++u) s += xv[u].x + xv[u].y + xv[u].z + xv[u].w;
        }
    }
    if (s == 123.456f) out[0] = s;
}

// the k-split pattern of k_dense0_fwd3: wave = (net, split s of NS, 128-column tile), 16-row steps s, s + NS, ...
template <int NS>
__global__ __launch_bounds__(256) void k_split(const float* __restrict__ w, float* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, bl = lane & 31, h = lane >> 5;
    int item = blockIdx.x * 4 + wave;
    const int jt = item % 4; item /= 4;
    const int s_ = item % NS, net = item / NS;
    if (net >= NETS) return;
    const int NU = F / 16, NC = (NU - s_ + NS - 1) / NS;
    const float* p = w + ((long)net * F + 16 * s_ + 8 * h) * J + jt * 128 + 4 * bl;
    float s = 0.f;
    for (int c = 0; c + 4 <= NC; c += 4) {
        f32x4v v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[u][jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p + ((long)(c + u) * 16 * NS + jj) * J));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) s += v[u][jj].x + v[u][jj].y + v[u][jj].z + v[u][jj].w;
    }
    if (s == 123.456f) out[0] = s;
}

template <typename Fl, typename Fn>
void run(const char* name, Fl flush, Fn launch, double bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) { flush(); launch(); }
    float best = 1e9f, sum = 0.f;
    for (int i = 0; i < 10; ++i) {
        flush(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("%-58s best %6.1f us = %.2f TB/s   mean %6.1f us\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12, sum / 10 * 1e3);
}

int main() {
    const size_t nw = (size_t)NETS * F * J;
    float *w, *x, *out, *dirty;
    hipMalloc(&w, nw * 4); hipMalloc(&x, (size_t)NETS * F * 24 * 4 + 65536); hipMalloc(&out, 256); hipMalloc(&dirty, 512u << 20);
    hipMemset(w, 0x11, nw * 4); hipMemset(x, 0x11, (size_t)NETS * F * 24 * 4 + 65536);
    const double bytes = (double)nw * 4;
    auto flush = [&]() { hipMemsetAsync(dirty, 1, 512u << 20, 0); };  // 512 MB of writes in front of every launch: nothing of W in any cache
    run("k-split  250 x 4 waves (k_dense0_fwd3's pattern, NS 25)", flush, [&]() { hipLaunchKernelGGL(k_split<25>, dim3(250), dim3(256), 0, 0, w, out); }, bytes);
    run("k-split  500 x 4 waves (NS 50)", flush, [&]() { hipLaunchKernelGGL(k_split<50>, dim3(500), dim3(256), 0, 0, w, out); }, bytes);
    run("col-tile 160 x  8 waves", flush, [&]() { hipLaunchKernelGGL((k_col<8, false>), dim3(160), dim3(512), 0, 0, w, x, out); }, bytes);
    run("col-tile 160 x 16 waves", flush, [&]() { hipLaunchKernelGGL((k_col<16, false>), dim3(160), dim3(1024), 0, 0, w, x, out); }, bytes);
    run("col-tile 160 x  8 waves + activation fragments from L2", flush, [&]() { hipLaunchKernelGGL((k_col<8, true>), dim3(160), dim3(512), 0, 0, w, x, out); }, bytes);
    run("col-tile 160 x 16 waves + activation fragments from L2", flush, [&]() { hipLaunchKernelGGL((k_col<16, true>), dim3(160), dim3(1024), 0, 0, w, x, out); }, bytes);
    return 0;
}
