// Probe (not part of the library): how fast can a CU fill LDS from L2-resident data?
// Each workgroup (256 threads) repeatedly copies CHUNK bytes from a small per-XCD-resident buffer into LDS,
// keeping DEPTH chunks in flight, with (a) LDS-DMA (global_load_lds 16 B/lane) or (b) register loads + ds_write.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/lds_fill_probe.hip -o /tmp/lds_fill_probe && /tmp/lds_fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// CHUNK_KB per stage, DEPTH stages in flight, ring of DEPTH+1 buffers
template <int CHUNK_KB, int DEPTH>
__global__ __launch_bounds__(256) void k_dma(const float* src, long src_floats, int iters, float* sink) {
    constexpr int CH = CHUNK_KB * 256;  // floats per chunk
    constexpr int NI = CHUNK_KB / 4;    // glds16 per wave... per thread-issue: 256 lanes x 16 B = 4 KB per WG-wide call
    __shared__ __attribute__((aligned(16))) float lds[(DEPTH + 1) * CH];
    const int t = threadIdx.x, wave = t >> 6;
    const float* base = src;
    long pos = ((long)blockIdx.x * CH) % (src_floats - CH);
    float acc = 0.f;
#define STAGE(i)                                                                                   \
    {                                                                                              \
        const float* g = base + pos + t * 4;                                                       \
        float* l = &lds[((i) % (DEPTH + 1)) * CH + wave * 256];                                    \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) glds16(g + j * 1024, l + j * 1024);         \
        pos += CH; if (pos + CH > src_floats) pos = 0;                                             \
    }
    for (int i = 0; i < DEPTH; ++i) STAGE(i)
    for (int i = 0; i < iters; ++i) {
        // wait for the oldest of the DEPTH chunks in flight
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");
        __builtin_amdgcn_s_barrier();
        STAGE(i + DEPTH)
        acc += lds[(i % (DEPTH + 1)) * CH + t];  // touch the landed chunk
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) sink[0] = acc;
}

template <int CHUNK_KB, int DEPTH>
__global__ __launch_bounds__(256) void k_reg(const float* src, long src_floats, int iters, float* sink) {
    constexpr int CH = CHUNK_KB * 256, NI = CHUNK_KB / 4;
    __shared__ __attribute__((aligned(16))) float lds[2 * CH];
    const int t = threadIdx.x;
    long pos = ((long)blockIdx.x * CH) % (src_floats - CH);
    float acc = 0.f;
    float4 r[DEPTH][NI];
#define LOADR(s)                                                                                  \
    {                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) r[s][j] = *reinterpret_cast<const float4*>(src + pos + t * 4 + j * 1024); \
        pos += CH; if (pos + CH > src_floats) pos = 0;                                            \
    }
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) LOADR(s)
    for (int i = 0; i < iters; i += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
#pragma unroll
            for (int j = 0; j < NI; ++j) *reinterpret_cast<float4*>(&lds[((i + s) & 1) * CH + t * 4 + j * 1024]) = r[s][j];
            LOADR(s)
            __syncthreads();
            acc += lds[((i + s) & 1) * CH + ((t * 7) & (CH - 1))];
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <typename F>
static double run(F launch, int iters, int n_wg, int chunk_kb) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)iters * n_wg * chunk_kb * 1024.0 / (ms * 1e-3) / 1e9;  // GB/s chip-wide
}

int main() {
    const long n_alloc = 8L << 20;  // 32 MB
    float *src, *sink;
    hipMalloc(&src, n_alloc * 4); hipMalloc(&sink, 4);
    hipMemset(src, 0, n_alloc * 4);
    const int iters = 2000;
    for (long n : {256L << 10, 8L << 20})  // 1 MB window (L2-resident) and 32 MB (Infinity Cache)
    for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {
        if (wg_per_cu == 1) printf("---- source window %ld MB\n", n * 4 >> 20);
        const int n_wg = 256 * wg_per_cu;
#define RUN_DMA(KB, D) printf("dma  chunk %2d KB depth %d  wg/cu %d : %7.0f GB/s chip  %6.1f GB/s/CU\n", KB, D, wg_per_cu, \
        run([&] { hipLaunchKernelGGL((k_dma<KB, D>), dim3(n_wg), dim3(256), 0, 0, src, n, iters, sink); }, iters, n_wg, KB), \
        run([&] { hipLaunchKernelGGL((k_dma<KB, D>), dim3(n_wg), dim3(256), 0, 0, src, n, iters, sink); }, iters, n_wg, KB) / 256);
#define RUN_REG(KB, D) printf("reg  chunk %2d KB depth %d  wg/cu %d : %7.0f GB/s chip  %6.1f GB/s/CU\n", KB, D, wg_per_cu, \
        run([&] { hipLaunchKernelGGL((k_reg<KB, D>), dim3(n_wg), dim3(256), 0, 0, src, n, iters, sink); }, iters, n_wg, KB), \
        run([&] { hipLaunchKernelGGL((k_reg<KB, D>), dim3(n_wg), dim3(256), 0, 0, src, n, iters, sink); }, iters, n_wg, KB) / 256);
        RUN_DMA(8, 1) RUN_DMA(8, 2) RUN_DMA(8, 3)
        RUN_DMA(16, 1) RUN_DMA(16, 2) RUN_DMA(16, 3)
        RUN_DMA(24, 1) RUN_DMA(24, 2)
        RUN_REG(8, 1) RUN_REG(8, 2) RUN_REG(16, 1) RUN_REG(16, 2)
    }
    return 0;
}
