// Can a once-read weight stream (the Dense_0 forward: 158.6 MB per step) be pulled faster through LDS-DMA than through
// vector registers?  MI355X_MICROARCH.md quotes ~10 B/clk/CU for global_load_dwordx4 and 12-13 B/clk/CU / 6.4-6.8 TB/s for
// LDS-DMA fills; the step's k_dense0_fwd3 reaches 4.4-4.6 TB/s through registers.  Same bytes, same grid shapes:
//   reg     256-thread workgroups, 4 x float4 in flight per thread, default or non-temporal loads (what k_dense0_fwd3 does)
//   dma     512-thread workgroups whose waves only issue global_load_lds_dwordx4 into a 64 / 128 KB LDS ring, throttled by a
//           counted s_waitcnt vmcnt (no consumer: an upper bound for a loader-ring kernel), default or nt
// each in two cache states: clean (after a 512 MB read of another buffer) and dirty (after 600 MB of writes).
// hipcc --offload-arch=gfx950 -O3 ldsdma_stream_probe.hip -o ldsdma_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_reg(const float* __restrict__ p, long n4, float* out) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    const f32x4v* q = reinterpret_cast<const f32x4v*>(p);
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(q + i + u * stride) : q[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 123.456f) out[0] = s;
}

template <bool NT>
__device__ __forceinline__ void dma16(unsigned voff, unsigned long sbase, unsigned lds_addr) {
    if (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// every wave: a contiguous run of 1 KiB pieces, grid-stride over (workgroup, wave); INFL pieces in flight per wave
template <bool NT, int INFL>
__global__ __launch_bounds__(512) void k_dma(const float* __restrict__ p, long n_pieces, int ring_pieces) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&lds[0];
    const long wid = (long)blockIdx.x * nw + wave, nwaves = (long)gridDim.x * nw;
    const int slots = ring_pieces / nw;  // ring slots of this wave
    int slot = 0, infl = 0;
    for (long piece = wid; piece < n_pieces; piece += nwaves) {
        dma16<NT>(lane * 16, (unsigned long)p + (unsigned long)piece * 1024, lds0 + (wave * slots + slot) * 1024);
        slot = slot + 1 == slots ? 0 : slot + 1;
        if (++infl == INFL) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL - 4) : "memory"); infl = INFL - 4; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void k_write(float4* q, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) q[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const long bytes = 158597120L / 1024 * 1024, n4 = bytes / 16, n_pieces = bytes / 1024;
    float *p, *other, *out;
    float4* wbuf;
    hipMalloc(&p, bytes); hipMalloc(&other, 512L << 20); hipMalloc(&wbuf, 600L << 20); hipMalloc(&out, 4);
    hipMemset(p, 1, bytes); hipMemset(other, 1, 512L << 20);
    hipFuncSetAttribute((const void*)k_dma<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k_dma<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k_dma<true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dirty = 0; dirty < 2; ++dirty)
        for (int mode = 0; mode < 8; ++mode) {
            float best = 1e9, sum = 0;
            const int reps = 7;
            for (int rep = 0; rep < reps; ++rep) {
                if (dirty) hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, wbuf, (600L << 20) / 16);
                else hipLaunchKernelGGL(k_reg<false>, dim3(1024), dim3(256), 0, 0, other, (512L << 20) / 16, out);
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_reg<false>, dim3(250), dim3(256), 0, 0, p, n4, out); break;
                    case 1: hipLaunchKernelGGL(k_reg<true>, dim3(250), dim3(256), 0, 0, p, n4, out); break;
                    case 2: hipLaunchKernelGGL(k_reg<true>, dim3(500), dim3(256), 0, 0, p, n4, out); break;
                    case 3: hipLaunchKernelGGL(k_reg<true>, dim3(1024), dim3(256), 0, 0, p, n4, out); break;
                    case 4: hipLaunchKernelGGL((k_dma<false, 16>), dim3(256), dim3(512), 65536, 0, p, n_pieces, 64); break;
                    case 5: hipLaunchKernelGGL((k_dma<true, 16>), dim3(256), dim3(512), 65536, 0, p, n_pieces, 64); break;
                    case 6: hipLaunchKernelGGL((k_dma<true, 16>), dim3(256), dim3(256), 65536, 0, p, n_pieces, 64); break;
                    case 7: hipLaunchKernelGGL((k_dma<true, 32>), dim3(256), dim3(512), 131072, 0, p, n_pieces, 128); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
                if (rep >= 2) sum += ms;
            }
            static const char* nm[8] = {"reg default, 250 x 256", "reg nt,      250 x 256", "reg nt,      500 x 256", "reg nt,     1024 x 256",
                                        "dma default, 256 x 8 waves, 16 in flight", "dma nt,      256 x 8 waves, 16 in flight",
                                        "dma nt,      256 x 4 waves, 16 in flight", "dma nt,      256 x 8 waves, 32 in flight"};
            printf("%s  %-44s best %.1f us = %.2f TB/s   mean %.1f us = %.2f TB/s\n", dirty ? "behind 600 MB of writes" : "clean caches           ", nm[mode],
                   best * 1e3, bytes / best / 1e9, sum / (reps - 2) * 1e3, bytes / (sum / (reps - 2)) / 1e9);
        }
    return 0;
}
