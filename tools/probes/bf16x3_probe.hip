// Probe (not part of the library): f32-accurate products on the bf16 matrix cores.
//   C[m][n] = sum_k A[k][m] * B[k][n],  A, B given as f32, split EXACTLY into three bf16 terms each
//   (8 + 8 + 8 significand bits), stored in LDS as rows [k][plane 0..2][32] of 16-bit elements (192 B per row),
//   fragments fetched with ds_read_b64_tr_b16 and multiplied by six v_mfma_f32_32x32x16_bf16 per 16-k step
//   (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0: everything down to 2^-24 of the product).
// Prints the worst error against a float64 reference beside that of a plain f32 fma chain.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/bf16x3_probe.hip -o /tmp/bf16x3_probe && /tmp/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
#define KDIM 64

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned int u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h0, unsigned short& h1, unsigned short& h2) {
    h0 = bf16_rne(x);
    float r = x - bf16_f(h0);
    h1 = bf16_rne(r);
    r = r - bf16_f(h1);
    h2 = bf16_rne(r);
}
__device__ __forceinline__ bf16x8 frag(const unsigned short* rows, int k0, int plane, int lane) {
    const int row = k0 + 8 * (lane >> 5) + ((lane & 15) >> 2), col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto p = (__attribute__((address_space(3))) s16x4*)(rows + row * 96 + plane * 32 + col);
    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p + 96);  // 4 rows further: 4 * 96 u16 = 96 s16x4
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, v0), __builtin_bit_cast(bf16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ void k_probe(const float* A, const float* B, float* C, float* residual) {
    __shared__ __attribute__((aligned(16))) unsigned short a3[KDIM * 96], b3[KDIM * 96];
    const int l = threadIdx.x;
    float worst = 0.f;
    for (int e = l; e < KDIM * 32; e += 64) {
        unsigned short h0, h1, h2;
        split3(A[e], h0, h1, h2);
        worst = fmaxf(worst, fabsf(A[e] - ((bf16_f(h0) + bf16_f(h1)) + bf16_f(h2))));
        a3[(e >> 5) * 96 + (e & 31)] = h0; a3[(e >> 5) * 96 + 32 + (e & 31)] = h1; a3[(e >> 5) * 96 + 64 + (e & 31)] = h2;
        split3(B[e], h0, h1, h2);
        b3[(e >> 5) * 96 + (e & 31)] = h0; b3[(e >> 5) * 96 + 32 + (e & 31)] = h1; b3[(e >> 5) * 96 + 64 + (e & 31)] = h2;
    }
    atomicMax((int*)residual, __float_as_int(worst));
    __syncthreads();
    f32x16 c = {0};
    for (int k0 = 0; k0 < KDIM; k0 += 16) {
        bf16x8 a0 = frag(a3, k0, 0, l), a1 = frag(a3, k0, 1, l), a2 = frag(a3, k0, 2, l);
        bf16x8 b0 = frag(b3, k0, 0, l), b1 = frag(b3, k0, 1, l), b2 = frag(b3, k0, 2, l);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
    }
    // D layout of the 32x32 MFMAs: register r of lane l = row (r&3) + 8 (r>>2) + 4 (l>>5), column l & 31
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
    std::vector<float> A(KDIM * 32), B(KDIM * 32), C(32 * 32);
    srand(7);
    for (auto& x : A) x = (float)((rand() / (double)RAND_MAX - 0.5) * exp((rand() % 12) - 6));
    for (auto& x : B) x = (float)((rand() / (double)RAND_MAX - 0.5) * exp((rand() % 12) - 6));
    float *dA, *dB, *dC, *dR, res = 0.f;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dR, 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dR, 0, 4);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dR);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(&res, dR, 4, hipMemcpyDeviceToHost);
    double worst3 = 0, worst32 = 0, scale = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            double ref = 0, mag = 0;
            float f = 0.f;
            for (int k = 0; k < KDIM; ++k) {
                ref += (double)A[k * 32 + m] * B[k * 32 + n];
                mag += fabs((double)A[k * 32 + m] * B[k * 32 + n]);
                f = fmaf(A[k * 32 + m], B[k * 32 + n], f);
            }
            worst3 = fmax(worst3, fabs(C[m * 32 + n] - ref) / mag);
            worst32 = fmax(worst32, fabs((double)f - ref) / mag);
            scale = fmax(scale, mag);
        }
    printf("three-term split residual (must be 0): %g\n", res);
    printf("worst |err| / sum|a b|:  bf16x3 (6 products) %.3e   plain f32 fma chain %.3e   (2^-24 = %.3e)\n", worst3,
           worst32, ldexp(1.0, -24));
    return (res == 0.f && worst3 <= 1.5 * worst32) ? 0 : 2;  // as accurate as the f32 chain
}
