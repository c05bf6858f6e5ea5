// Single-wave issue rate of v_mfma_f32_32x32x16_bf16 by the number of independent accumulators between two products of the
// same tile, with and without LDS fragment reads in the stream (the i-IQN GEMM k-step: DESIGN 3.3b).  One wave per SIMD
// (256-thread workgroups, one per CU), or two (512 threads).  Cycles per MFMA from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_probe.hip -o /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CH, bool LDS>
__global__ void k(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 48 * 1024 / 4; i += blockDim.x) ((unsigned*)lds)[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u + (unsigned)lane, 0x3f803f80u});
    bf16x8 b = a;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 48 / CH; ++rep) {
            if (LDS) {  // one fragment read per CH products (the GEMM: 18 reads per 48 products)
                a = *(const __attribute__((address_space(3))) bf16x8*)(lds + ((it * 7 + rep) & 31) * 1024 + lane * 16);
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CH, bool LDS>
void run(int threads, const char* what) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<CH, LDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[256 * 8];
    (void)hipMemcpy(h, cyc, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 256 * (threads / 64);
    for (int i = 0; i < n; ++i) s += (double)h[i];
    printf("%-28s %d waves/SIMD  %2d accumulators in turn: %.1f cycles per MFMA per wave (%.1f per SIMD slot)\n", what, threads / 256, CH,
           s / n / (iters * 48.0), s / n / (iters * 48.0) / (threads / 256));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int threads : {256, 512}) {
        run<1, false>(threads, "registers only"); run<2, false>(threads, "registers only"); run<4, false>(threads, "registers only");
        run<2, true>(threads, "one ds_read_b128 per group"); run<4, true>(threads, "one ds_read_b128 per group");
    }
    return 0;
}
