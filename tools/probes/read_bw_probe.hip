// Pure-read HBM bandwidth probe: how fast can MI355X stream a buffer that is only READ (the Dense_0 forward reads
// 158.6 MB of weights per step and never writes them).  hipcc --offload-arch=gfx950 -O3 read_bw_probe.hip -o read_bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
template <int U>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, long n4, float* out) {
    // grid-stride over U independent float4 per thread and trip
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 123.456f) out[0] = s;
}
template <int U>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ p, float4* __restrict__ q, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) q[i + u * stride] = v[u];
    }
}
int main() {
    const long bytes = getenv("MB") ? atol(getenv("MB")) << 20 : 1L << 30, n4 = bytes / 16;
    float4 *p, *q; float* out;
    hipMalloc(&p, bytes); hipMalloc(&q, bytes); hipMalloc(&out, 4);
    hipMemset(p, 1, bytes); hipMemset(q, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_read<4>, dim3(wgs), dim3(256), 0, 0, p, n4, out);
                if (mode == 1) hipLaunchKernelGGL(k_read<8>, dim3(wgs), dim3(256), 0, 0, p, n4, out);
                if (mode == 2) hipLaunchKernelGGL(k_copy<4>, dim3(wgs), dim3(256), 0, 0, p, q, n4);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%5d WGs  %-8s  %.3f ms  %.2f TB/s %s\n", wgs, mode == 0 ? "read x4" : mode == 1 ? "read x8" : "copy x4", best,
                   (mode == 2 ? 2.0 : 1.0) * bytes / best / 1e9, mode == 2 ? "(read + write)" : "");
        }
    }
    return 0;
}
